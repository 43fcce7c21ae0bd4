#!/usr/bin/env python3
"""Parity + step time of the dtype policies on the committed full-size fixtures: fp16 everywhere, bf16 everywhere, and the mix
"tower bf16 + LLM fp16".   python tools/mixed_dtype_step.py [case ...]"""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoitg_amd import synth
from videoitg_amd.config import ITGConfig
from videoitg_amd.engine import SelectorEngine
from videoitg_amd.synth_device import packed_random_weights

cases = sys.argv[1:] or ["c2_512x448"]
cfg = ITGConfig()
dev = torch.device("cuda:0")
F, B = torch.float16, torch.bfloat16
recs = {c: json.load(open(os.path.join(ROOT, "tests", "golden", f"full_{c}.json"))) for c in cases
        if os.path.exists(os.path.join(ROOT, "tests", "golden", f"full_{c}.json"))}
for name, (ldt, vdt) in {"fp16": (F, F), "bf16": (B, B), "tower bf16 + llm fp16": (F, B)}.items():
    eng, eng_seed = None, None
    for case, rec in recs.items():
        if eng_seed != rec["weight_seed"]:                       # fixtures with other weight seeds get their own model
            if eng is not None:
                eng.release_workspace()
            eng, eng_seed = None, rec["weight_seed"]
            torch.cuda.empty_cache()
            eng = SelectorEngine(packed_random_weights(cfg, ldt, dev, seed=eng_seed, vit_dtype=vdt))
        ids, pix = synth.workload_inputs(cfg, rec["frames"], rec["resolution"], rec["prompt_len"], rec["pixel_seed"])
        pix = pix.to(vdt).to(dev)
        want = torch.tensor(rec["logits"])
        lg = eng.forward(ids, None, [pix])[0].view(-1).float().cpu()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            eng.forward(ids, None, [pix])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 * 1e3
        ds = (torch.sigmoid(lg) - torch.sigmoid(want)).abs().max().item()
        top = sorted(torch.sort(torch.sigmoid(lg), descending=True, stable=True).indices[:rec["k"]].tolist()) == rec["topk"]
        print(f"{name:24s} {case:12s}: max|dscore| {ds:.2e}  top{rec['k']} match {top}  (gap {rec['gap_k_k1']})  {ms:.1f} ms/step", flush=True)
    eng.release_workspace()
    del eng
    torch.cuda.empty_cache()
