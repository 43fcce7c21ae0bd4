#!/usr/bin/env python3
"""Writes tests/golden/heavy_tail_plan.json: the per-layer norm-gain pairs of videoitg_amd.synth.HeavyTailPlan.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/calibrate_heavy_tail.py

One complete fp32 oracle pass over the full-width, full-depth model on a SMALL input (4 frames of 224^2, the 60-token prompt; ~3 min on
8 vCPUs), layer by layer: before each layer the residual stream is measured (scale seen by the norm over ALL channels, scale of the
ordinary channels alone, mean magnitude of the planted channels) and the layer's two norm gains get the power-of-two pair that hands
the following projections ordinary channels of unit scale and planted channels of ~4x that - what a trained checkpoint's gains do for
its massive-activation channels.  The pairs are data: every heavy-tailed fixture (make_fullsize_golden.py cases ending in "o") and
the engine on the GPU box rebuild the same weights from the record the fixture carries.
"""
import json
import math
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import videoitg_oracle as O  # noqa: E402
from videoitg_amd import synth  # noqa: E402
from videoitg_amd.config import ITGConfig  # noqa: E402

PLANTED_POST_NORM = 4.0


def pow2(v: float) -> float:
    return 2.0 ** round(math.log2(max(v, 1e-30)))


def gain_pair(x: torch.Tensor, chans, centred: bool):
    """x [..., d] -> (ordinary gain, planted gain, stats) as powers of two"""
    d = x.shape[-1]
    keep = torch.ones(d, dtype=torch.bool)
    keep[list(chans)] = False
    xf = x.reshape(-1, d).double()
    if centred:
        xf = xf - xf.mean(-1, keepdim=True)
    s_all = xf.pow(2).mean(-1).sqrt()                        # what the norm divides by, per token
    s_ord = xf[:, keep].pow(2).mean(-1).sqrt()
    a = xf[:, ~keep].abs().mean()
    g_ord = pow2(float((s_all / s_ord).mean()))
    g_out = pow2(PLANTED_POST_NORM * float(s_all.mean()) / max(float(a), 1e-6))
    return (g_ord, g_out), dict(norm_scale=float(s_all.mean()), ordinary_rms=float(s_ord.mean()), planted_abs=float(a),
                                absmax=float(x.abs().max()))


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = ITGConfig()
    v = cfg.vision
    T, R, L = 4, 224, 60
    ids, pix = synth.workload_inputs(cfg, T, R, L)
    vc, lc = synth.HeavyTailPlan.channels(cfg)
    n_vit = cfg.num_vit_layers_needed()
    plan = synth.HeavyTailPlan(vc, lc, vit_norm=[], llm_norm=[], final_norm=(1.0, 1.0))
    vspecs, lspecs = list(synth.vision_param_specs(cfg)), list(synth.llm_param_specs(cfg))
    log = []
    with torch.inference_mode():
        x = O.siglip_embed(pix.float(), synth.counter_state_dict([s for s in vspecs if s[0].startswith("vision_model.embeddings.")], 0), v.patch_size)
        for i in range(n_vit):
            pair, st = gain_pair(x, vc, centred=True)
            plan.vit_norm.append(pair)
            sd = synth.counter_state_dict([s for s in vspecs if s[0].startswith(f"vision_model.encoder.layers.{i}.")], 0, plan=plan)
            x = O.siglip_layer(x, sd, i, v.num_attention_heads, v.layer_norm_eps)
            log.append(dict(stage=f"vit{i}", gains=pair, **st))
            print(log[-1], flush=True)
        print("tower out absmax", float(x.abs().max()), flush=True)
        sd = synth.counter_state_dict([s for s in lspecs if not s[0].startswith("model.layers.") and s[0] != "model.embed_tokens.weight"], 0, plan=plan)
        visual = O.mlp_projector(x, sd, cfg.vision_token_num)
        text = ids.reshape(-1)[ids.reshape(-1) != O.IMAGE_TOKEN_INDEX]
        espec = next(s for s in lspecs if s[0] == "model.embed_tokens.weight")
        small = plan.apply({"model.embed_tokens.weight": synth.counter_rows(0, espec[0], espec[1], text, espec[2], espec[3])})["model.embed_tokens.weight"]
        remap = ids.reshape(-1).clone()
        remap[remap != O.IMAGE_TOKEN_INDEX] = torch.arange(text.numel())
        x, _, tts = O.splice(remap.unsqueeze(0), None, [visual], {"model.embed_tokens.weight": small},
                             max_length=cfg.tokenizer_model_max_length, padding_side=cfg.tokenizer_padding_side)
        cos, sin = O.rope_tables(x.shape[1], cfg.head_dim, cfg.rope_theta)
        for i in range(cfg.num_hidden_layers):
            pair, st = gain_pair(x, lc, centred=False)
            plan.llm_norm.append(pair)
            sdi = synth.counter_state_dict([s for s in lspecs if s[0].startswith(f"model.layers.{i}.")], 0, plan=plan)
            x = O.qwen2_layer(x, sdi, i, num_heads=cfg.num_attention_heads, num_kv_heads=cfg.num_key_value_heads,
                              eps=cfg.rms_norm_eps, cos=cos, sin=sin, key_mask=None)
            log.append(dict(stage=f"llm{i}", gains=pair, **st))
            print(log[-1], flush=True)
        pair, st = gain_pair(x, lc, centred=False)
        plan.final_norm = pair
        log.append(dict(stage="final", gains=pair, **st))
        print(log[-1], flush=True)
        sd = synth.counter_state_dict([s for s in lspecs if not s[0].startswith("model.layers.") and s[0] != "model.embed_tokens.weight"], 0, plan=plan)
        logits = O.score_head(O.rms_norm(x, sd["model.norm.weight"], cfg.rms_norm_eps), tts, [T], sd)[0].reshape(-1)
        print("logits", logits.tolist())
    rec = dict(plan=plan.to_json(), calibrated_on=dict(frames=T, resolution=R, prompt_len=L, weight_seed=0, planted_post_norm=PLANTED_POST_NORM),
               stream=log)
    with open(os.path.join(HERE, "heavy_tail_plan.json"), "w") as f:
        json.dump(rec, f)
    print("wrote heavy_tail_plan.json")


if __name__ == "__main__":
    main()
