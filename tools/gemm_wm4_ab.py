#!/usr/bin/env python3
"""Round 3: the 384x256 tile of gemm_mi16.hip as a 2 x 4 wave grid (192 x 64 wave tiles, shipped) against a 4 x 2 grid (96 x 128 wave tiles: 12.5 %
fewer fragment bytes per flop), the step's shapes with their real epilogues, interleaved in one process (median of rounds).  Also checks that the
two grids give bit-identical results (same k order per output element) and shows the 256x384 tile (128 x 96 wave tiles) where the mode has it.
    python tools/gemm_wm4_ab.py [fp16|bf16] [rounds]"""
import os, sys, statistics
os.environ.setdefault("VITG_PROFILING_LIB", "1")   # research knobs (round 4: the product library takes four option keys only)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoitg_amd import ops
dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float16
rounds = next((int(a) for a in sys.argv[1:] if a.isdigit()), 3)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [("vit_qkv +bias", 524288, 3456, 1152, 0, 0), ("vit_fc1 +bias+gelu", 524288, 4352, 1152, 0, ops.ACT_GELU_TANH),
          ("llm_qkv +bias", 12860, 4608, 3584, 0, 0), ("llm_o +res", 12860, 3584, 3584, 1, 0),
          ("llm_gu swiglu", 12860, 37888, 3584, 2, 0), ("llm_down +res", 12860, 3584, 18944, 1, 0), ("8192^3", 8192, 8192, 8192, 0, 0),
          ("vit_out +res16", 524288, 1152, 1152, 3, 0), ("vit_fc2 +res16", 524288, 1152, 4352, 3, 0)]
for name, M, N, K, mode, act in shapes:
    a = torch.randn((M, K), device="cuda").to(dt)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    bias = None if mode == 2 or ("res" in name and "vit" not in name) or name == "8192^3" else torch.randn(N, device="cuda") * 0.02
    res0 = torch.randn((M, N), device="cuda") if mode == 1 else None
    out = torch.zeros((M, N // 2 if mode == 2 else N), dtype=torch.float32 if mode == 1 else dt, device="cuda")
    x0 = torch.randn((M, N), device="cuda").to(dt) if mode == 3 else None
    def fn():
        if mode == 3:
            return ops.gemm_res16(a, w, out, bias=bias, res16=x0)
        return ops.gemm(a, w, out, mode=mode, bias=bias, act=act, res=res0)
    variants = [("384x256 2x4", 10, -1), ("384x256 4x2", 10, 15)] + ([("256x384", 9, -1)] if mode != 2 else []) + [("auto", 0, -1)]
    t = {v[0]: [] for v in variants}
    ref, diff = None, 0.0
    for r in range(rounds):
        for vn, c, wm4 in variants:
            ops.set_option("gemm_cfg", c); ops.set_option("gemm_wm4", wm4)
            t[vn].append(timeit(fn))
            if r == 0 and c == 10:
                if ref is None: ref = out[:4096].float().clone()
                else: diff = max(diff, (out[:4096].float() - ref).abs().max().item())
    ops.set_option("gemm_cfg", 0); ops.set_option("gemm_wm4", 0)
    fl = 2.0 * M * N * K / 1e9
    print(f"{name:20s} {M:7d}x{N:5d}x{K:5d}: " + "   ".join(f"{vn} {fl / statistics.median(t[vn]):6.0f}" for vn, _, _ in variants) +
          f"   TF/s   4x2 vs 2x4 {statistics.median(t['384x256 2x4']) / statistics.median(t['384x256 4x2']) - 1:+.1%}   max|diff| {diff:.2e}", flush=True)
    del a, w, out, res0, x0
