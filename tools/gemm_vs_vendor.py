#!/usr/bin/env python3
"""This library's GEMM against the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the shapes of the step, same box, same call,
random fp16 / bf16 operands, plain 16-bit output without epilogue (what both sides can do).  A sanity check of where the kernel stands:
the vendor kernels run under the same 1400 W package power cap (profiles/r02_gemm_timeline.md).
    python tools/gemm_vs_vendor.py [fp16|bf16]"""
import os, sys, torch
os.environ.setdefault("TORCH_BLAS_PREFER_HIPBLASLT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops

dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float16


def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"torch {torch.__version__}, preferred BLAS: {torch.backends.cuda.preferred_blas_library()}", flush=True)
for name, M, N, K in [("8192^3", 8192, 8192, 8192), ("vit_qkv", 524288, 3456, 1152), ("vit_fc1", 524288, 4352, 1152), ("vit_fc2 (no residual)", 524288, 1152, 4352),
                      ("llm_qkv", 12860, 4608, 3584), ("llm_gate/up (no SwiGLU)", 12860, 37888, 3584), ("llm_down (no residual)", 12860, 3584, 18944)]:
    a = torch.randn((M, K), device="cuda").to(dt)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    out = torch.empty((M, N), dtype=dt, device="cuda")
    wt = w.t()
    t_v = timeit(lambda: torch.matmul(a, wt, out=out))
    ref = out.float().clone() if M * N <= 8192 * 8192 else None
    t_o = timeit(lambda: ops.gemm(a, w, out, mode=0))
    err = (out.float() - ref).abs().max().item() if ref is not None else float("nan")
    fl = 2.0 * M * N * K / 1e9
    print(f"{name:26s} {M:7d} x {N:5d} x {K:5d}: vendor {fl / t_v:7.1f} TF/s ({t_v:.3f} ms)   this library {fl / t_o:7.1f} TF/s ({t_o:.3f} ms)   "
          f"ratio {t_v / t_o:.2f}   max|diff| {err:.3g}", flush=True)
    del a, w, out, wt, ref
