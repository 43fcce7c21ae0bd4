#!/usr/bin/env python3
"""GPU: vitg_gemm_fp8 (fp32-output form) on heavy-column operands against the float64 product of the dequantised operands - how far, where, relative to what"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops
DEV = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
for (M, N, K, scale, heavy) in [(1720, 856, 128, 30.0, 50.0), (1720, 856, 128, 1.0, 1.0), (24, 1336, 2816, 30.0, 50.0), (31, 1896, 768, 1.0, 50.0), (512, 512, 1152, 1.0, 1.0)]:
    x = torch.randn((M, K), generator=g) * scale
    x[:, 7] *= heavy
    w = torch.randn((N, K), generator=g) * K ** -0.5
    a8, sa = ops.quant_rows_fp8(x.to(DEV))
    w8, sw = ops.quant_rows_fp8(w.to(DEV))
    aq = a8[:, :K].cpu().view(torch.float8_e4m3fn).double()
    wq = w8[:, :K].cpu().view(torch.float8_e4m3fn).double()
    acc = aq @ wq.t()                                                        # exact integer-ish products in float64
    ref = acc * sa.cpu().double()[:, None] * sw.cpu().double()[None, :]
    out = torch.zeros((M, N), device=DEV)
    ops.gemm_fp8(a8, sa, w8, sw, out, mode=ops.GEMM_F32, out_dtype=torch.float16)
    got = out.cpu().double()
    err = (got - ref).abs()
    i = int(err.argmax())
    m, n = divmod(i, N)
    sumabs = (aq.abs() @ wq.abs().t())
    print(f"M={M} N={N} K={K} scale={scale} heavy={heavy}: max err {err.max():.3e} at ({m},{n}) ref {ref[m, n]:.6e} got {got[m, n]:.6e} rel {err.max() / abs(ref[m, n]):.2e} | "
          f"|acc| there {abs(acc[m, n]):.3e} sum|terms| there {sumabs[m, n]:.3e} -> err / (sa sw sum|terms|) = {err.max() / (sumabs[m, n] * sa[m].item() * sw[n].item()):.2e} | "
          f"max rel err over outputs with |ref| > 1e-3 max: {(err / ref.abs().clamp(min=1e-3 * ref.abs().max())).max():.2e}")
