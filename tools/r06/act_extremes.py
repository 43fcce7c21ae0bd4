#!/usr/bin/env python3
"""GPU: the GEMM epilogues' activations on pre-activation values far outside what random operands produce (+-300, +-1e4, +-6e4, bf16 +-1e30, tiny, -0):
finite where torch is finite, equal to torch's fp32 GELU (tanh / erf) / SiLU * up after one 16-bit rounding"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops
DEV = torch.device("cuda:0")
bad = 0
for dt in (torch.float16, torch.bfloat16):
    vals = torch.cat([torch.linspace(-300, 300, 1201), torch.tensor([1e4, -1e4, 6e4, -6e4, 1e-7, -1e-7, 0.0, -0.0, 5.5, -5.5, 9.0, -9.0, 12.0, -12.0, 40.0, -40.0])])
    if dt == torch.bfloat16:
        vals = torch.cat([vals, torch.tensor([1e30, -1e30, 3e38, -3e38, 1e-30])])
    vals = vals.to(dt).float()
    M, K, N = vals.numel(), 64, 64
    a = torch.zeros((M, K)); a[:, 0] = vals
    w = torch.zeros((N, K)); w[:, 0] = 1.0
    for act, fn, name in ((ops.ACT_GELU_TANH, lambda x: F.gelu(x, approximate="tanh"), "gelu_tanh"), (ops.ACT_GELU_ERF, F.gelu, "gelu_erf")):
        out = torch.empty((M, N), dtype=dt, device=DEV)
        ops.gemm(a.to(dt).to(DEV), w.to(dt).to(DEV), out, mode=ops.GEMM_STORE16, act=act)
        got = out[:, 0].float().cpu()
        want = fn(vals).to(dt).float()                                      # torch fp32, rounded once
        fin = torch.isfinite(want)
        nonfinite = int((~torch.isfinite(got[fin])).sum())
        err = ((got[fin] - want[fin]).abs() / want[fin].abs().clamp(min=1e-3)).max().item()
        worst = vals[fin][((got[fin] - want[fin]).abs() / want[fin].abs().clamp(min=1e-3)).argmax()].item()
        # where torch itself overflows (its erf form computes x * (1 + erf) before the 0.5: inf at 3e38) the kernel may return the mathematical value x
        same_inf = all(bool(g_ == w_) or (torch.isinf(w_) and bool(g_ == v_)) for g_, w_, v_ in zip(got[~fin], want[~fin], vals[~fin])) if (~fin).any() else True
        print(f"{name} {dt}: non-finite where torch is finite: {nonfinite}, max rel err {err:.2e} (at x = {worst}), overflow cases equal: {same_inf}")
        bad += nonfinite + (err > 2.0 ** (-7 if dt == torch.bfloat16 else -10)) + (not same_inf)
    # SwiGLU: gate = v, up = 1 -> silu(v)
    g = torch.zeros((32, K)); g[:, 0] = 1.0
    u = torch.zeros((32, K)); u[:, 1] = 1.0
    a2 = a.clone(); a2[:, 1] = 1.0
    gu = torch.cat([g, u], 0)
    out = torch.empty((M, 64), dtype=dt, device=DEV)
    ops.gemm(a2.to(dt).to(DEV), gu.to(dt).to(DEV), out, mode=ops.GEMM_SWIGLU)
    got = out[:, 0].float().cpu()
    want = F.silu(vals).to(dt).float()
    fin = torch.isfinite(want)
    nonfinite = int((~torch.isfinite(got[fin])).sum())
    err = ((got[fin] - want[fin]).abs() / want[fin].abs().clamp(min=1e-3)).max().item()
    print(f"swiglu {dt}: non-finite where torch is finite: {nonfinite}, max rel err {err:.2e}")
    bad += nonfinite + (err > 2.0 ** (-7 if dt == torch.bfloat16 else -10))
print("act extremes:", "clean" if not bad else f"{bad} problems")
sys.exit(1 if bad else 0)
