#!/usr/bin/env python3
"""CPU (build container): what do 16-bit MFMA operands cost on the heavy-tailed fixture, kernels aside?

oracle/quant_model.py restates the forward with roundings inserted (a) exactly where libvideoitg_hip.so rounds ("product": MFMA inputs only, fp32
streams / accumulators / statistics) and (b) after EVERY op, as the reference's own `model.half()` inference does ("reference16",
/root/reference/eagle/model/builder.py:59, infer.py:21).  Both are run on the full-width, full-depth model with the heavy-tailed weights of
tests/golden/full_depth16o_224.json (16 frames x 224^2) and with the plain draw (full_depth16_224.json), against the committed fp32 oracle logits.
If an independent fp32-accumulating CPU model with the product's rounding sites lands where the HIP path lands, the fixture's error is the price of
16-bit operands on weights with massive-activation channels - and "reference16" says what the reference's own arithmetic pays on the same weights.

    PYTHONDONTWRITEBYTECODE=1 python tools/heavy_tail_rounding_model.py [case ...]      (~30 GB of RAM; default cases ~25 min on 8 vCPUs, c2o_512x448 ~90 min)"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import quant_model as QM            # noqa: E402
from videoitg_amd import synth                  # noqa: E402
from videoitg_amd.config import ITGConfig       # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = ITGConfig()
    print("| fixture | rounding model | max abs dlogit | rms dlogit | max abs dscore |\n|---|---|---|---|---|", flush=True)
    cases = sys.argv[1:] or ["depth16o_224", "depth16_224"]
    for case in cases:
        rec = json.load(open(os.path.join(ROOT, "tests", "golden", f"full_{case}.json")))
        plan = synth.fixture_plan(rec)
        ids, pix = synth.fixture_inputs(cfg, rec)
        n_vit = cfg.num_vit_layers_needed()
        vsd = synth.counter_state_dict([s for s in synth.vision_param_specs(cfg) if s[0].startswith("vision_model.embeddings.") or int(s[0].split(".")[3]) < n_vit],
                                       rec["weight_seed"], plan=plan)
        lspecs = list(synth.llm_param_specs(cfg))
        sd = synth.counter_state_dict([s for s in lspecs if s[0] != "model.embed_tokens.weight"], rec["weight_seed"], plan=plan)
        # the prompt's embedding rows only (the table is 545 M values), re-indexed like oracle/fullsize.py does
        flat = ids.reshape(-1)
        text = flat[flat != -200]
        espec = next(s for s in lspecs if s[0] == "model.embed_tokens.weight")
        small = synth.counter_rows(rec["weight_seed"], espec[0], espec[1], text, espec[2], espec[3])
        if plan is not None:
            small = plan.apply({"model.embed_tokens.weight": small})["model.embed_tokens.weight"]
        sd["model.embed_tokens.weight"] = small
        remap = flat.clone()
        remap[flat != -200] = torch.arange(text.numel())
        want = torch.tensor(rec["logits"])
        big = rec["frames"] > 64                       # the headline size: 45 min per model on 8 vCPUs - the two fp16 models only
        for mode, dt in (("product", torch.float16), ("reference16", torch.float16)) + (() if big else (("product", torch.bfloat16),)):
            with torch.inference_mode():
                got = QM.forward(cfg, sd, vsd, remap.unsqueeze(0), None, [pix], dtype=dt, mode=mode, frame_chunk=16)[0].view(-1).float()
            dl = got - want
            print(f"| {case} | {mode} {str(dt).split('.')[-1]} | {dl.abs().max().item():.2e} | {dl.pow(2).mean().sqrt().item():.2e} | "
                  f"{(torch.sigmoid(got) - torch.sigmoid(want)).abs().max().item():.2e} |", flush=True)
        del sd, vsd


if __name__ == "__main__":
    main()
