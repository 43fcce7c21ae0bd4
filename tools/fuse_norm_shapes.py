#!/usr/bin/env python3
"""Cost of the fused-norm producer / consumer forms against the plain GEMMs + the stand-alone norm they replace (same call, fp16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dt = torch.float16
M = 524288
for name, N, K in (("vit_out", 1152, 1152), ("vit_fc2", 1152, 4352)):
    a = torch.randn((M, K), device="cuda").to(dt); w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    bias = torch.randn(N, device="cuda"); gamma = torch.ones(N, device="cuda"); x = torch.randn((M, N), device="cuda")
    xh = torch.empty((M, N), dtype=dt, device="cuda"); h = torch.empty((M, N), dtype=dt, device="cuda")
    cols = ops.gemm_norm_plan(M, N, K); P = N // cols
    part = torch.empty(M * P * 2, device="cuda"); stats = torch.empty(M * 2, device="cuda")
    t_plain = timeit(lambda: ops.gemm(a, w, x, mode=ops.GEMM_F32, bias=bias, res=x))
    t_prod = timeit(lambda: ops.gemm_norm_producer(a, w, x, res=x, next_gamma=gamma, xh=xh, partials=part, bias=bias))
    t_fin = timeit(lambda: ops.norm_stats_finalize(part, M, P, cols, 1e-6, False, stats))
    t_ln = timeit(lambda: ops.layernorm(x, gamma, bias, h, N, 1e-6))
    print(f"{name}: plain {t_plain:.3f} ms + layernorm {t_ln:.3f} ms = {t_plain + t_ln:.3f} | producer {t_prod:.3f} + finalize {t_fin:.3f} = {t_prod + t_fin:.3f} ms (cols {cols})", flush=True)
    del a, w, x, xh, h, part
for name, N, K, act in (("vit_qkv", 3456, 1152, 0), ("vit_fc1", 4352, 1152, 1)):
    a = torch.randn((M, K), device="cuda").to(dt); w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    bias = torch.randn(N, device="cuda"); cs = torch.randn(N, device="cuda"); out = torch.empty((M, N), dtype=dt, device="cuda")
    stats = torch.rand(M * 2, device="cuda")
    t_plain = timeit(lambda: ops.gemm(a, w, out, mode=ops.GEMM_STORE16, bias=bias, act=act))
    t_cons = timeit(lambda: ops.gemm_norm_consumer(a, w, out, mode=ops.GEMM_STORE16, stats=stats, colsum=cs, bias=bias, act=act))
    print(f"{name}: plain {t_plain:.3f} ms | consumer {t_cons:.3f} ms", flush=True)
    del a, w, out
