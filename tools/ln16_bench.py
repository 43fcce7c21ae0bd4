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
print(f"lib={os.path.basename(os.environ.get('VITG_LIB_PATH', 'product'))} cap={os.environ.get('VITG_LN16_CAP', 'default')} block={os.environ.get('VITG_LN16_BLOCK', 'default')}: {ms * 1e3:.1f} us  {M * D * 4 / ms / 1e9:.2f} TB/s")
if "--check" in sys.argv:            # against torch's fp32 LayerNorm of the same 16-bit rows, and this box's plain copy rate for the same bytes
    want = torch.nn.functional.layer_norm(x[:8192].float(), (D,), w, b, 1e-6)
    print(f"max |diff| vs fp32 LayerNorm (8192 rows): {(out[:8192].float() - want).abs().max().item():.3e}  (one 16-bit rounding of values up to {want.abs().max().item():.1f})")
    for _ in range(3): out.copy_(x)
    e0.record()
    for _ in range(20): out.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"torch copy of the same bytes: {ms * 1e3:.1f} us  {M * D * 4 / ms / 1e9:.2f} TB/s")
