#!/usr/bin/env python3
"""fp8 weight path on one part only: parity vs the complete fp32 oracle pass and step time (python tools/fp8_parts.py [case])."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoitg_amd import synth
from videoitg_amd.config import ITGConfig
from videoitg_amd.engine import SelectorEngine
from videoitg_amd.synth_device import packed_random_weights
case = sys.argv[1] if len(sys.argv) > 1 else "c2_512x448"
rec = json.load(open(os.path.join(ROOT, "tests", "golden", f"full_{case}.json")))
cfg = ITGConfig(); dev = torch.device("cuda:0")
ids, pix = synth.workload_inputs(cfg, rec["frames"], rec["resolution"], rec["prompt_len"], rec["pixel_seed"])
want = torch.tensor(rec["logits"]); pix = pix.half().to(dev)
for fp8 in (False, "vit", "llm", True):
    eng = SelectorEngine(packed_random_weights(cfg, torch.float16, dev, seed=0, fp8=fp8))
    lg = eng.forward(ids, None, [pix])[0].view(-1).float().cpu()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): eng.forward(ids, None, [pix])
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 2 * 1e3
    ds = (torch.sigmoid(lg) - torch.sigmoid(want)).abs().max().item()
    top = set(torch.sort(torch.sigmoid(lg), descending=True, stable=True).indices[:rec["k"]].tolist())
    print(f"fp8={str(fp8):5s} (fp16 elsewhere): max|dscore| {ds:.2e}, top{rec['k']} overlap {len(top & set(rec['topk']))}/{rec['k']}, {ms:.1f} ms/step", flush=True)
    eng.release_workspace(); del eng; torch.cuda.empty_cache()
