#!/usr/bin/env python3
"""GPU: does the fp16 MFMA keep SUBNORMAL inputs?  The operand-split tier's lo = fp16(x - fp16(x)) is subnormal whenever |x| < 2^-3 (|lo| <= 2^-12 |x| < 2^-14).
A GEMM whose A operand has |a| ~ 1e-2 (every lo subnormal) against unit-scale weights: error relative to fp64 ~2^-22 if subnormals enter the product,
~2^-12 if the matrix pipe flushes them."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd.torch_ops import K

dev = "cuda:0"
torch.manual_seed(0)
for scale in (1.0, 0.05, 0.01, 1e-3):
    M, N, Kd = 512, 512, 256
    a = torch.randn(M, Kd, device=dev) * scale
    w = torch.randn(N, Kd, device=dev).half()
    hi = a.half()
    lo = (a - hi.float()).half()
    a2 = torch.cat([hi, lo], 1).contiguous()
    out = torch.zeros(M, N, device=dev)
    K.gemm(a2, torch.cat([w, w], 1).contiguous(), out, mode=K.GEMM_F32)
    ref = (hi.double() + lo.double()) @ w.double().t()
    ref32 = a.double() @ w.double().t()
    only_hi = hi.double() @ w.double().t()
    den = ref32.abs().mean()
    sub = (lo.abs() < 2.0 ** -14).float().mean().item()
    print(f"|a| ~ {scale:g}: lo subnormal {sub:.0%}; rel err vs (hi+lo) {((out.double() - ref).abs().mean() / den).item():.2e}, vs fp32 a {((out.double() - ref32).abs().mean() / den).item():.2e}; "
          f"hi only would be {((only_hi - ref32).abs().mean() / den).item():.2e}")
