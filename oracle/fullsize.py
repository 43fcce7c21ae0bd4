"""ORACLE — test infrastructure, NOT product code.

Streaming driver of ``oracle/videoitg_oracle.py`` for the full-size architecture: the same oracle functions, called layer by
layer with weights produced on the fly by ``videoitg_amd.synth.counter_tensor`` (bit-identical on CPU and GPU), so a complete
fp32 pass over the 7.6 B-parameter model fits a 62 GB host (the 30 GB of fp32 LLM weights never exist at once).

Used by ``tests/golden/make_fullsize_golden.py`` (which commits the resulting logits as fixtures for the `-m gpu` parity
tests and for ``bench.py``'s ``parity`` block) and by ``tests/test_fullsize_oracle.py`` (streaming == ``O.forward`` on a small
configuration).  Reference call sites are the ones cited in ``videoitg_oracle.py``; this file adds no arithmetic of its own.
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import torch

from oracle import videoitg_oracle as O
from videoitg_amd import synth


def _specs(it, pred):
    return [s for s in it if pred(s[0])]


def workload_inputs(cfg, T: int, R: int, prompt_len: int, pix_seed: int = 1234):
    """the synthetic inputs of SURVEY.md §8(d); ONE function (videoitg_amd.synth.workload_inputs) feeds bench.py, the GPU parity
    tests and this oracle"""
    return synth.workload_inputs(cfg, T, R, prompt_len, pix_seed)


def streaming_forward(cfg, input_ids: torch.Tensor, pix: torch.Tensor, seed: int = 0, frame_chunk: int = 16,
                      log=None, plan=None) -> Dict[str, object]:
    """One sample (B=1, one <image>), counter-generated weights of `cfg`'s architecture -> dict(logits fp32 [T], checks, seconds)."""
    say = log or (lambda *_: None)
    v = cfg.vision
    T = pix.shape[0]
    t0 = time.perf_counter()
    n_vit = cfg.num_vit_layers_needed()
    vsd = synth.counter_state_dict(
        _specs(synth.vision_param_specs(cfg),
               lambda k: k.startswith("vision_model.embeddings.") or int(k.split(".")[3]) < n_vit), seed, plan=plan)
    t_w = time.perf_counter() - t0
    t1 = time.perf_counter()
    feats = O.siglip_tower(pix.float(), vsd, patch=v.patch_size, num_heads=v.num_attention_heads,
                           num_layers=v.num_hidden_layers, select_layer=cfg.mm_vision_select_layer, eps=v.layer_norm_eps,
                           frame_chunk=frame_chunk)
    t_vit = time.perf_counter() - t1
    say(f"tower done in {t_vit:.0f}s")
    del vsd
    lspecs = list(synth.llm_param_specs(cfg))
    t2 = time.perf_counter()
    sd = synth.counter_state_dict(_specs(lspecs, lambda k: not k.startswith("model.layers.") and k != "model.embed_tokens.weight"), seed, plan=plan)
    t_w += time.perf_counter() - t2
    visual = O.mlp_projector(feats, sd, cfg.vision_token_num)
    checks = dict(vit_sum=float(feats.double().sum()), vit_abs=float(feats.double().abs().sum()),
                  vit_sample=feats.reshape(-1)[:: max(1, feats.numel() // 16)][:16].tolist(),
                  visual_sum=float(visual.double().sum()), visual_abs=float(visual.double().abs().sum()),
                  vit_absmax=float(feats.abs().max()), visual_absmax=float(visual.abs().max()))
    del feats
    # embedding rows of the prompt only (the table is 545 M values; 60 rows are read)
    ids = input_ids.reshape(-1)
    text = ids[ids != O.IMAGE_TOKEN_INDEX]
    espec = next(s for s in lspecs if s[0] == "model.embed_tokens.weight")
    small = synth.counter_rows(seed, espec[0], espec[1], text, espec[2], espec[3])
    if plan is not None:
        small = plan.apply({"model.embed_tokens.weight": small})["model.embed_tokens.weight"]
    remap = ids.clone()
    remap[ids != O.IMAGE_TOKEN_INDEX] = torch.arange(text.numel())
    x, mask, tts = O.splice(remap.unsqueeze(0), None, [visual], {"model.embed_tokens.weight": small},
                            max_length=cfg.tokenizer_model_max_length, padding_side=cfg.tokenizer_padding_side)
    del visual
    N = x.shape[1]
    cos, sin = O.rope_tables(N, cfg.head_dim, cfg.rope_theta)
    t_llm = 0.0
    layer_abs, layer_max = [], []
    for i in range(cfg.num_hidden_layers):
        t3 = time.perf_counter()
        sdi = synth.counter_state_dict(_specs(lspecs, lambda k: k.startswith(f"model.layers.{i}.")), seed, plan=plan)
        t_w += time.perf_counter() - t3
        t4 = time.perf_counter()
        x = O.qwen2_layer(x, sdi, i, num_heads=cfg.num_attention_heads, num_kv_heads=cfg.num_key_value_heads,
                          eps=cfg.rms_norm_eps, cos=cos, sin=sin, key_mask=None)
        t_llm += time.perf_counter() - t4
        layer_abs.append(float(x.double().abs().mean()))
        layer_max.append(float(x.abs().max()))
        del sdi
        say(f"llm layer {i} done ({t_llm:.0f}s so far)")
    h = O.rms_norm(x, sd["model.norm.weight"], cfg.rms_norm_eps)
    logits = O.score_head(h, tts, [T], sd)[0].reshape(-1)
    checks["llm_layer_mean_abs"] = layer_abs
    checks["llm_absmax"] = max(layer_max)
    checks["final_sum"] = float(h.double().sum())
    return dict(logits=logits, N=N, checks=checks,
                seconds=dict(weights=round(t_w, 1), tower=round(t_vit, 1), llm=round(t_llm, 1),
                             total=round(time.perf_counter() - t0, 1)))


def ranking_summary(logits: torch.Tensor, k: int = 32) -> Dict[str, object]:
    """A13 on the oracle logits + the rank-k / rank-k+1 score gap BASELINE.md §3 asks to report next to a Top-k match."""
    scores, order = O.rank_frames(logits)
    k = min(k, logits.numel())
    gap = float(scores[k - 1] - scores[k]) if logits.numel() > k else None
    return dict(order=order.tolist(), topk=sorted(order[:k].tolist()), k=k, gap_k_k1=gap,
                scores=[float(s) for s in torch.sigmoid(logits.float()).tolist()])
