#!/usr/bin/env python3
"""Error of each attention loop structure against the committed golden logits (tests/golden/*.npz), fp16."""
import os, sys
os.environ.setdefault("VITG_PROFILING_LIB", "1")   # research knobs (round 4: the product library takes four option keys only)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from videoitg_amd import ops
from test_gpu_e2e import run_case
gd = os.path.join(ROOT, "tests", "golden")
for name in ["pool_post", "nopool_prepost", "oddgrid", "batch2_pad", "chunk520"]:
    res = {}
    for sp, label in [(2, "per-tile"), (1, "sw-pipelined"), (0, "default")]:
        ops.set_option("attn_sp", sp)
        z, got, T = run_case(gd, name, torch.float16)
        dl = ds = 0.0
        for b in range(len(T)):
            ref = torch.from_numpy(z[f"logits{b}"]).view(-1)
            g = got[b].view(-1)
            dl = max(dl, (g - ref).abs().max().item())
            ds = max(ds, (torch.sigmoid(g) - torch.sigmoid(ref)).abs().max().item())
        res[label] = (dl, ds)
    ops.set_option("attn_sp", 0)
    print(name, "  ".join(f"{k}: |dlogit| {v[0]:.2e} |dscore| {v[1]:.2e}" for k, v in res.items()), flush=True)
