#!/usr/bin/env python3
"""Where does the heavy-tailed fixture's error enter?  (round 5: full_c2o_512x448 / full_depth16o_224 sit ~30x above the benign fixtures in logit error.)

Full-width, full-depth model on a SMALL input (default 16 frames x 224^2, the depth16o case), heavy-tailed weights (synth.HeavyTailPlan from
tests/golden/heavy_tail_plan.json): the fp32 oracle streamed layer by layer on the host cores (checker side; ~1-2 min) with its intermediates kept,
the HIP engine with the same weights, compared stage by stage - tower stream after selected layers, tower output, projected visual rows, decoder
stream after k layers (the engine run with its decoder truncated to k layers so that its in-place stream can be read), logits.  Errors are reported
separately for the PLANTED channels and the ordinary ones, relative to each group's own rms.  Then the same comparison for the plain draw (no plan)
as the baseline.     python tools/heavy_tail_stage_diff.py [frames=16] [res=224]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import videoitg_oracle as O          # noqa: E402  (tools/: checker side)
from videoitg_amd import synth                   # noqa: E402
from videoitg_amd.config import ITGConfig        # noqa: E402
from videoitg_amd.engine import SelectorEngine   # noqa: E402
from videoitg_amd.synth_device import packed_random_weights   # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R = int(sys.argv[2]) if len(sys.argv) > 2 else 224
LLM_STAGES = (0, 1, 2, 4, 8, 16, 28)
VIT_STAGES = (1, 3, 8, 16, 26)


def grp(a, b, chans):
    """(rms err / rms, max abs err) over ordinary and planted channels of [rows, d] tensors"""
    a, b = a.float().cpu(), b.float().cpu()
    d = b.shape[-1]
    keep = torch.ones(d, dtype=torch.bool)
    if chans:
        keep[list(chans)] = False
    out = []
    for m in (keep, ~keep):
        if m.sum() == 0:
            out.append("-")
            continue
        e, r = (a[:, m] - b[:, m]), b[:, m]
        out.append(f"{(e.pow(2).mean().sqrt() / r.pow(2).mean().sqrt().clamp(min=1e-30)).item():.2e} (max {e.abs().max().item():.2e} on rms {r.pow(2).mean().sqrt().item():.2e})")
    return out


def oracle_pass(cfg, ids, pix, plan):
    v = cfg.vision
    vspecs, lspecs = list(synth.vision_param_specs(cfg)), list(synth.llm_param_specs(cfg))
    n_vit = cfg.num_vit_layers_needed()
    inter = {}
    x = O.siglip_embed(pix.float(), synth.counter_state_dict([s for s in vspecs if s[0].startswith("vision_model.embeddings.")], 0, plan=plan), v.patch_size)
    for i in range(n_vit):
        sd = synth.counter_state_dict([s for s in vspecs if s[0].startswith(f"vision_model.encoder.layers.{i}.")], 0, plan=plan)
        x = O.siglip_layer(x, sd, i, v.num_attention_heads, v.layer_norm_eps)
        if i + 1 in VIT_STAGES:
            inter[f"vit{i + 1}"] = x.clone()
    sd = synth.counter_state_dict([s for s in lspecs if not s[0].startswith("model.layers.") and s[0] != "model.embed_tokens.weight"], 0, plan=plan)
    visual = O.mlp_projector(x, sd, cfg.vision_token_num)
    inter["visual"] = visual.clone()
    flat = ids.reshape(-1)
    text = flat[flat != O.IMAGE_TOKEN_INDEX]
    espec = next(s for s in lspecs if s[0] == "model.embed_tokens.weight")
    small = synth.counter_rows(0, espec[0], espec[1], text, espec[2], espec[3])
    if plan is not None:
        small = plan.apply({"model.embed_tokens.weight": small})["model.embed_tokens.weight"]
    remap = flat.clone()
    remap[flat != O.IMAGE_TOKEN_INDEX] = torch.arange(text.numel())
    xs, _, tts = O.splice(remap.unsqueeze(0), None, [visual], {"model.embed_tokens.weight": small}, max_length=None, padding_side="right")
    inter["llm0"] = xs[0].clone()
    cos, sin = O.rope_tables(xs.shape[1], cfg.head_dim, cfg.rope_theta)
    for i in range(cfg.num_hidden_layers):
        sdi = synth.counter_state_dict([s for s in lspecs if s[0].startswith(f"model.layers.{i}.")], 0, plan=plan)
        xs = O.qwen2_layer(xs, sdi, i, num_heads=cfg.num_attention_heads, num_kv_heads=cfg.num_key_value_heads, eps=cfg.rms_norm_eps, cos=cos, sin=sin,
                           key_mask=None)
        if i + 1 in LLM_STAGES:
            inter[f"llm{i + 1}"] = xs[0].clone()
    h = O.rms_norm(xs, sd["model.norm.weight"], cfg.rms_norm_eps)
    inter["logits"] = O.score_head(h, tts, [pix.shape[0]], sd)[0].reshape(-1)
    return inter


def main():
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    cfg = ITGConfig()
    dev = torch.device("cuda:0")
    ids, pix = synth.workload_inputs(cfg, T, R, 60)
    heavy = synth.HeavyTailPlan.from_json(json.load(open(os.path.join(ROOT, "tests", "golden", "heavy_tail_plan.json")))["plan"])
    for tag, plan in (("heavy-tailed", heavy), ("plain draw", None)):
        with torch.inference_mode():
            ref = oracle_pass(cfg, ids, pix, plan)
        pw = packed_random_weights(cfg, torch.float16, dev, seed=0, plan=plan)
        eng = SelectorEngine(pw)
        eng.skip_dead_rows = False
        vc, lc = (heavy.vit_channels, heavy.llm_channels)
        print(f"\n## {tag}: {T} frames x {R}^2; rms err / rms of the group (max abs err on the group's rms)\n| stage | ordinary channels | planted channels |\n|---|---|---|")
        full_vit = list(pw.vit)
        G = R // cfg.vision.patch_size
        hw = cfg.pooled_grid(T, G)
        for r16 in (True, False):
            eng.residual16 = r16
            for k in VIT_STAGES:
                pw.vit = full_vit[:k]
                feats = eng.vit_forward(pix.half().to(dev), None)[0]
                torch.cuda.synchronize()
                o, p = grp(feats.view(-1, feats.shape[-1])[:, :cfg.mm_hidden_size], ref[f"vit{k}"].reshape(-1, cfg.mm_hidden_size), vc)
                print(f"| tower stream after {k} layer(s), residual stream {'16-bit' if r16 else 'fp32'} | {o} | {p} |")
        pw.vit = full_vit
        eng.residual16 = True
        full_llm = list(pw.llm)
        n = ref["llm0"].shape[0]
        for k in LLM_STAGES:
            pw.llm = full_llm[:k]
            eng.forward(ids, None, [pix.half().to(dev)])
            torch.cuda.synchronize()
            x = eng._ws["llm_x"][:n]
            o, p = grp(x, ref[f"llm{k}"], lc)
            print(f"| {'spliced input (projector out + text)' if k == 0 else f'decoder stream after {k} layer(s)'} | {o} | {p} |")
        pw.llm = full_llm
        got = eng.forward(ids, None, [pix.half().to(dev)])[0].view(-1).float().cpu()
        want = ref["logits"]
        print(f"| logits | max abs err {(got - want).abs().max().item():.2e}, rms err {(got - want).pow(2).mean().sqrt().item():.2e} on a spread (std) of "
              f"{want.std().item():.2f} | scores: max abs err {(torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item():.2e} |", flush=True)
        eng.release_workspace()
        del eng, pw
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
