#!/usr/bin/env python3
"""Tile-shape sweep of the shipped 256-tile GEMM on the step's shapes WITH their real epilogues (gemm_cfg 0 = auto, 6 = 256x256,
9 = 256x384, 10 = 384x256), same call: checks that auto mode still picks the fastest shape.   python tools/gemm_cfg_sweep.py [fp16|bf16]"""
import os, sys, torch
os.environ.setdefault("VITG_PROFILING_LIB", "1")   # research knobs (round 4: the product library takes four option keys only)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops
dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float16

def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

shapes = [("vit_qkv +bias", 524288, 3456, 1152, 0, 0), ("vit_fc1 +bias+gelu", 524288, 4352, 1152, 0, ops.ACT_GELU_TANH), ("vit_out +res", 524288, 1152, 1152, 1, 0),
          ("vit_fc2 +res", 524288, 1152, 4352, 1, 0), ("llm_qkv +bias", 12860, 4608, 3584, 0, 0), ("llm_o +res", 12860, 3584, 3584, 1, 0),
          ("llm_gu swiglu", 12860, 37888, 3584, 2, 0), ("llm_down +res", 12860, 3584, 18944, 1, 0)]
for name, M, N, K, mode, act in shapes:
    a = torch.randn((M, K), device="cuda").to(dt)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    bias = None if mode == 2 or "res" in name and "vit" not in name else torch.randn(N, device="cuda") * 0.02
    out = torch.zeros((M, N // 2 if mode == 2 else N), dtype=torch.float32 if mode == 1 else dt, device="cuda")
    row = []
    for cfg in (0, 6, 9, 10):
        if cfg == 9 and mode == 2: row.append("   -  "); continue
        ops.set_option("gemm_cfg", cfg)
        t = timeit(lambda: ops.gemm(a, w, out, mode=mode, bias=bias, act=act, res=out if mode == 1 else None))
        row.append(f"{2.0 * M * N * K / t / 1e9:6.0f}")
    ops.set_option("gemm_cfg", 0)
    print(f"{name:20s} {M:7d}x{N:5d}x{K:5d}: auto {row[0]}   256x256 {row[1]}   256x384 {row[2]}   384x256 {row[3]}  TF/s", flush=True)
    del a, w, out
