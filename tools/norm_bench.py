#!/usr/bin/env python3
"""LayerNorm / RMSNorm achieved HBM bandwidth on the production shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops
for name, M, D, rms in [("vit_ln448", 524288, 1152, False), ("vit_ln384", 373248, 1152, False), ("llm_rms", 12860, 3584, True)]:
    x = torch.randn((M, D), device="cuda")
    w, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    out = torch.empty((M, D), dtype=torch.float16, device="cuda")
    f = (lambda: ops.rmsnorm(x, w, out, D, 1e-6)) if rms else (lambda: ops.layernorm(x, w, b, out, D, 1e-6))
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:10s} M={M} D={D}: {ms:.3f} ms  {M * D * 6 / ms / 1e9:.2f} TB/s (4 B read + 2 B write per element)")
