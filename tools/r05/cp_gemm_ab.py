#!/usr/bin/env python3
"""One context-parallel rank's decoder GEMMs, per shape: the 320 x 256 tile (cfg 12, round 5) against the tiles it replaces, same call.
   gate/up (SwiGLU) forced onto cfg 10 (384 x 256) / 12 (320 x 256) / auto; o and down through split-K x2 (256-row tiles) / x3 (320-row tiles)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops

def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

D, F = 3584, 18944
for M in (1600, 1660):
    a = (torch.rand((M, D), device="cuda") * 2 - 1).half()
    gu = ((torch.rand((2 * F, D), device="cuda") * 2 - 1) * D ** -0.5).half()
    f = torch.empty((M, F), dtype=torch.float16, device="cuda")
    row = [f"M={M} gate/up:"]
    for cfg in (0, 10, 12, 6, 0):
        ops.set_option("gemm_cfg", cfg)
        row.append(f"cfg{cfg} {t(lambda: ops.gemm(a, gu, f, mode=ops.GEMM_SWIGLU)):.0f}us")
    ops.set_option("gemm_cfg", 0)
    print(" ".join(row), flush=True)
    for name, N, K in (("o", D, D), ("down", D, F)):
        aa = (torch.rand((M, K), device="cuda") * 2 - 1).half()
        w = ((torch.rand((N, K), device="cuda") * 2 - 1) * K ** -0.5).half()
        x = torch.zeros((M, N), dtype=torch.float32, device="cuda")
        ws = torch.empty(4 * M * N, dtype=torch.float32, device="cuda")
        row = [f"M={M} {name} (K={K}): plan {ops.gemm_splitk_plan(M, N, K)[0]};"]
        for sp in (2, 3, 4, 2, 3):
            row.append(f"x{sp} {t(lambda: ops.gemm_splitk(aa, w, x, ws, sp, res=x)):.0f}us")
        row.append(f"plain {t(lambda: ops.gemm(aa, w, x, mode=ops.GEMM_F32, res=x)):.0f}us")
        print(" ".join(row), flush=True)
