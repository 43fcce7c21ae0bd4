#!/usr/bin/env python3
"""GEMM throughput at the per-rank row counts of the context-parallel LLM (M = N_tokens / world)."""
import os, sys, torch
os.environ.setdefault("VITG_PROFILING_LIB", "1")   # research knobs (round 4: the product library takes four option keys only)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops
dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float16
for M in (1660, 3260, 6460):
    for name, N, K, mode in [("qkv", 4608, 3584, ops.GEMM_STORE16), ("o", 3584, 3584, ops.GEMM_F32), ("gateup", 37888, 3584, ops.GEMM_SWIGLU),
                             ("down", 3584, 18944, ops.GEMM_F32)]:
        a = (torch.rand((M, K), device="cuda") * 2 - 1).to(dt)
        w = ((torch.rand((N, K), device="cuda") * 2 - 1) * K ** -0.5).to(dt)
        if mode == ops.GEMM_F32:
            out = torch.zeros((M, N), dtype=torch.float32, device="cuda"); kw = dict(res=out)
        elif mode == ops.GEMM_SWIGLU:
            out = torch.empty((M, N // 2), dtype=dt, device="cuda"); kw = {}
        else:
            out = torch.empty((M, N), dtype=dt, device="cuda"); kw = {}
        res = []
        for tile, cfg in [(0, 0), (128, 0), (256, 6), (256, 2), (256, 9), (256, 10)]:
            ops.set_option("gemm_tile", tile); ops.set_option("gemm_cfg", cfg)
            for _ in range(3): ops.gemm(a, w, out, mode=mode, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm(a, w, out, mode=mode, **kw)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            res.append(f"{2.0 * M * N * K / ms / 1e9:6.0f}")
        ops.set_option("gemm_tile", 0); ops.set_option("gemm_cfg", 0)
        sk = ""
        if mode == ops.GEMM_F32:
            sp, nb = ops.gemm_splitk_plan(M, N, K)
            if sp > 1:
                ws = torch.empty(nb // 4, dtype=torch.float32, device="cuda")
                for _ in range(3): ops.gemm_splitk(a, w, out, ws, sp, res=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm_splitk(a, w, out, ws, sp, res=out)
                e1.record(); torch.cuda.synchronize()
                sk = f" | split-K x{sp} (incl. reduce) {2.0 * M * N * K / (e0.elapsed_time(e1) / 10) / 1e9:6.0f}"
        if mode == ops.GEMM_F32:
            nbk, wsb = ops.gemm_streamk_plan(M, N, K)
            if nbk > 0:
                ws = torch.empty(wsb // 4, dtype=torch.float32, device="cuda")
                for _ in range(3): ops.gemm_streamk(a, w, out, ws, nbk, res=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm_streamk(a, w, out, ws, nbk, res=out)
                e1.record(); torch.cuda.synchronize()
                sk += f" | stream-K {nbk} blocks (incl. reduce) {2.0 * M * N * K / (e0.elapsed_time(e1) / 10) / 1e9:6.0f}"
        print(f"M={M:5d} {name:7s} TF/s: auto {res[0]} | tile128 {res[1]} | cfg6 256x256 {res[2]} | cfg2 256x128 {res[3]} | cfg9 256x384 {res[4]} | cfg10 384x256 {res[5]}{sk}", flush=True)
