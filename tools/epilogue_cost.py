#!/usr/bin/env python3
"""What the activation arithmetic in the GEMM epilogue costs: fc1 shape with and without GELU, gate/up with SwiGLU vs a plain
16-bit store of the same width (same call, fp16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops

def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for dt in (torch.float16, torch.bfloat16):
    M, N, K = 524288, 4352, 1152
    a = torch.randn((M, K), device="cuda").to(dt); w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    bias = torch.randn(N, device="cuda"); out = torch.empty((M, N), dtype=dt, device="cuda")
    for rep in range(2):
        t0 = timeit(lambda: ops.gemm(a, w, out, mode=0, bias=bias))
        t1 = timeit(lambda: ops.gemm(a, w, out, mode=0, bias=bias, act=ops.ACT_GELU_TANH))
        t2 = timeit(lambda: ops.gemm(a, w, out, mode=0, bias=bias, act=ops.ACT_GELU_ERF))
        print(f"{dt} vit_fc1: bias only {t0:.3f} ms, + gelu_tanh {t1:.3f} ms (+{(t1 / t0 - 1) * 100:.1f} %), + gelu_erf {t2:.3f} ms (+{(t2 / t0 - 1) * 100:.1f} %)", flush=True)
    del a, w, out
    M, N, K = 12860, 37888, 3584
    a = torch.randn((M, K), device="cuda").to(dt); w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    o1 = torch.empty((M, N), dtype=dt, device="cuda"); o2 = torch.empty((M, N // 2), dtype=dt, device="cuda")
    for rep in range(2):
        t0 = timeit(lambda: ops.gemm(a, w, o1, mode=0))
        t1 = timeit(lambda: ops.gemm(a, w, o2, mode=2))
        print(f"{dt} llm_gate_up: plain 16-bit store (2x the bytes) {t0:.3f} ms, SwiGLU {t1:.3f} ms", flush=True)
    del a, w, o1, o2
