#!/usr/bin/env python3
"""Step time and parity with and without the fused norms (same engine, alternating): python tools/fuse_norm_step.py [fp16|bf16]"""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoitg_amd import synth
from videoitg_amd.config import ITGConfig
from videoitg_amd.engine import SelectorEngine
from videoitg_amd.synth_device import packed_random_weights
dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float16
rec = json.load(open(os.path.join(ROOT, "tests", "golden", "full_c2_512x448.json")))
cfg = ITGConfig(); dev = torch.device("cuda:0")
eng = SelectorEngine(packed_random_weights(cfg, dt, dev, seed=0))
ids, pix = synth.workload_inputs(cfg, 512, 448, 60, 1234)
pix = pix.to(dt).to(dev); want = torch.tensor(rec["logits"])
for rep in range(3):
    for fuse in (False, True):
        eng.fuse_norm = fuse
        lg = eng.forward(ids, None, [pix])[0].view(-1).float().cpu()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2): eng.forward(ids, None, [pix])
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 2 * 1e3
        ds = (torch.sigmoid(lg) - torch.sigmoid(want)).abs().max().item()
        top = sorted(torch.sort(torch.sigmoid(lg), descending=True, stable=True).indices[:32].tolist()) == rec["topk"]
        print(f"{dt} fuse_norm={fuse}: {ms:.1f} ms/step, max|dscore| vs oracle {ds:.2e}, top32 match {top}", flush=True)
