#!/usr/bin/env python3
"""Round 3: the shipped tile shapes on v_mfma_f32_32x32x16 (gemm256.hip) against v_mfma_f32_16x16x32 (gemm_mi16.hip), the step's shapes
with their real epilogues, interleaved in one process (median of rounds), product library.  Also prints max |difference| of the two
results (different accumulation order inside the MFMA: fp32 rounding only).
    python tools/gemm_mi_ab.py [fp16|bf16] [rounds]"""
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


shapes = [("vit_qkv +bias", 524288, 3456, 1152, 0, 0), ("vit_fc1 +bias+gelu", 524288, 4352, 1152, 0, ops.ACT_GELU_TANH), ("vit_out +res", 524288, 1152, 1152, 1, 0),
          ("vit_fc2 +res", 524288, 1152, 4352, 1, 0), ("llm_qkv +bias", 12860, 4608, 3584, 0, 0), ("llm_o +res", 12860, 3584, 3584, 1, 0),
          ("llm_gu swiglu", 12860, 37888, 3584, 2, 0), ("llm_down +res", 12860, 3584, 18944, 1, 0), ("8192^3", 8192, 8192, 8192, 0, 0),
          ("vit_out +res16", 524288, 1152, 1152, 3, 0), ("vit_fc2 +res16", 524288, 1152, 4352, 3, 0)]      # mode 3 = vitg_gemm_res16 (16x16x32 kernels only)
for name, M, N, K, mode, act in shapes:
    a = torch.randn((M, K), device="cuda").to(dt)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    bias = None if mode == 2 or ("res" in name and "vit" not in name) or name == "8192^3" else torch.randn(N, device="cuda") * 0.02
    res0 = torch.randn((M, N), device="cuda") if mode == 1 else None
    out = torch.zeros((M, N // 2 if mode == 2 else N), dtype=torch.float32 if mode == 1 else dt, device="cuda")
    fn = (lambda: ops.gemm_res16(a, w, out, bias=bias, res16=out)) if mode == 3 else (lambda: ops.gemm(a, w, out, mode=mode, bias=bias, act=act, res=res0))
    if mode == 3:
        out.normal_()
    cfgs = [0, 6, 10] + ([9] if mode != 2 else [])
    t = {(c, mi): [] for c in cfgs for mi in (32, 16)}
    diff = 0.0
    for r in range(rounds):
        for c in cfgs:
            ops.set_option("gemm_cfg", c)
            ref = None
            for mi in (32, 16):
                ops.set_option("gemm_mi", mi)
                t[(c, mi)].append(timeit(fn))
                if r == 0:
                    if ref is None: ref = out[:2048].float().clone()
                    else: diff = max(diff, (out[:2048].float() - ref).abs().max().item())
    ops.set_option("gemm_cfg", 0); ops.set_option("gemm_mi", 0)
    fl = 2.0 * M * N * K / 1e9
    cell = lambda c: f"{fl / statistics.median(t[(c, 32)]):6.0f} -> {fl / statistics.median(t[(c, 16)]):6.0f} ({statistics.median(t[(c, 32)]) / statistics.median(t[(c, 16)]) - 1:+.1%})"
    print(f"{name:20s} {M:7d}x{N:5d}x{K:5d}: " + "   ".join(f"{'auto' if c == 0 else {6: '256x256', 9: '256x384', 10: '384x256'}[c]} {cell(c)}" for c in cfgs) +
          f"   TF/s (32x32x16 -> 16x16x32)   max|diff| {diff:.2e}", flush=True)
    del a, w, out, res0
