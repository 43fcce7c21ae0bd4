#!/usr/bin/env python3
"""fp8 GEMM (vitg_gemm_fp8) vs the 16-bit GEMM on the shapes of the step: TF/s per shape, same call."""
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

shapes = [("8192^3", 8192, 8192, 8192, 0), ("vit_qkv", 524288, 3456, 1152, 0), ("vit_out", 524288, 1152, 1152, 1), ("vit_fc1", 524288, 4352, 1152, 0),
          ("vit_fc2", 524288, 1152, 4352, 1), ("llm_qkv", 12860, 4608, 3584, 0), ("llm_o", 12860, 3584, 3584, 1), ("llm_gu", 12860, 37888, 3584, 2),
          ("llm_down", 12860, 3584, 18944, 1)]
dt = torch.float16
for name, M, N, K, mode in shapes:
    a = (torch.randn((M, K), device="cuda") ).to(dt)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    out = torch.empty((M, N if mode != 2 else N // 2), dtype=torch.float32 if mode == 1 else dt, device="cuda")
    res = out if mode == 1 else None
    a8, sa = ops.quant_rows_fp8(a, K)
    w8, sw = ops.quant_rows_fp8(w, K)
    t16 = timeit(lambda: ops.gemm(a, w, out, mode=mode, res=res))
    t8 = timeit(lambda: ops.gemm_fp8(a8, sa, w8, sw, out, mode=mode, out_dtype=dt, res=res))
    tq = timeit(lambda: ops.quant_rows_fp8(a, K, a8, sa))
    fl = 2.0 * M * N * K / 1e9
    if name in ("vit_out", "vit_fc2"):        # the tower's residual GEMMs on its 16-bit stream (round 4): fp8 operands, 16-bit residual in / out
        x16 = torch.randn((M, N), device="cuda").to(dt)
        t8r = timeit(lambda: ops.gemm_fp8_res16(a8, w8, sw, x16, res16=x16, sa=sa))
        t16r = timeit(lambda: ops.gemm_res16(a, w, x16, res16=x16))
        print(f"{name:9s} 16-bit residual stream: fp16 {2.0 * M * N * K / 1e9 / t16r:7.1f} TF/s ({t16r:.3f} ms)   fp8 {2.0 * M * N * K / 1e9 / t8r:7.1f} TF/s ({t8r:.3f} ms)", flush=True)
        del x16
    print(f"{name:9s} M={M:6d} N={N:5d} K={K:5d} mode={mode}: fp16 {fl / t16:7.1f} TF/s ({t16:.3f} ms)   fp8 {fl / t8:7.1f} TF/s ({t8:.3f} ms)   "
          f"quant_rows(A) {tq:.3f} ms = {(M * K * 3) / tq / 1e9:.2f} TB/s", flush=True)
    del a, w, out, a8, w8
