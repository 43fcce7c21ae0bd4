#!/usr/bin/env python3
"""LayerNorm over 16-bit rows (vitg_layernorm16) at the tower's size: time and HBM rate (2 B read + 2 B written per element).
    VITG_LN16_CAP=<blocks> VITG_LN16_BLOCK=<threads> python tools/ln16_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops
M, D = 524288, 1152
x = torch.randn((M, D), device="cuda").half()
w, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
out = torch.empty((M, D), dtype=torch.float16, device="cuda")
for _ in range(3): ops.layernorm16(x, w, b, out, D, 1e-6)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.layernorm16(x, w, b, out, D, 1e-6)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"cap={os.environ.get('VITG_LN16_CAP', 'default')} block={os.environ.get('VITG_LN16_BLOCK', 'default')}: {ms * 1e3:.1f} us  {M * D * 4 / ms / 1e9:.2f} TB/s")
