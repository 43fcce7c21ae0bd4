#!/usr/bin/env python3
"""Per-shape GEMM throughput on the GPU (random fp16/bf16 data).  VITG_GEMM_TILE=128|256 pins the kernel.
    python tools/gemm_bench.py [fp16|bf16] [gemm_cfg=N ...]"""
import os
import sys
import torch

os.environ.setdefault('VITG_PROFILING', '1'); os.environ.setdefault('VITG_PROFILING_LIB', '1')   # lets gemm_flags ablation bits through (they produce wrong results)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops  # noqa: E402

dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float16
for arg in sys.argv[1:]:          # opt=value pairs go to vitg_set_option, e.g. gemm_cfg=8
    if "=" in arg:
        k_, v_ = arg.split("=")
        ops.set_option(k_, int(v_))
SHAPES = [  # name, M, N, K, mode
    ("vit_qkv", 524288, 3456, 1152, ops.GEMM_STORE16),
    ("vit_fc1", 524288, 4352, 1152, ops.GEMM_STORE16),
    ("vit_out", 524288, 1152, 1152, ops.GEMM_F32),
    ("vit_fc2", 524288, 1152, 4352, ops.GEMM_F32),
    ("llm_qkv", 12860, 4608, 3584, ops.GEMM_STORE16),
    ("llm_o", 12860, 3584, 3584, ops.GEMM_F32),
    ("llm_gateup", 12860, 37888, 3584, ops.GEMM_SWIGLU),
    ("llm_down", 12860, 3584, 18944, ops.GEMM_F32),
    ("square8k", 8192, 8192, 8192, ops.GEMM_STORE16),
]
for name, M, N, K, mode in SHAPES:
    a = (torch.rand((M, K), device="cuda") * 2 - 1).to(dt)
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) * K ** -0.5).to(dt)
    if "zeros" in sys.argv[1:]:       # data-dependent power: all-zero operands toggle no multiplier inputs
        a.zero_(); w.zero_()
    if "ones" in sys.argv[1:]:
        a.fill_(1.0); w.fill_(1.0)
    if mode == ops.GEMM_F32:
        out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
        kw = dict(res=out)
    elif mode == ops.GEMM_SWIGLU:
        out = torch.empty((M, N // 2), dtype=dt, device="cuda")
        kw = {}
    else:
        out = torch.empty((M, N), dtype=dt, device="cuda")
        kw = dict(bias=torch.zeros(N, device="cuda"), act=ops.ACT_GELU_TANH if "fc1" in name else ops.ACT_NONE)
    if "wf" in sys.argv[1:]:          # fragment-packed weights (gemm_wf.hip); first check against the LDS kernel
        ref = ops.gemm(a, w, out.clone(), mode=mode, **({**kw, "res": torch.zeros_like(out)} if "res" in kw else kw))
        kw["wf"] = ops.pack_weight_fragments(w)
        got = ops.gemm(a, w, out.clone(), mode=mode, **({**kw, "res": torch.zeros_like(out)} if "res" in kw else kw))
        d = (got.float() - ref.float()).abs().max().item()
        print(f"  wf vs lds: max|diff| {d:.3e}  ref max {ref.float().abs().max().item():.3e}", flush=True)
        del ref, got
    for _ in range(2):
        ops.gemm(a, w, out, mode=mode, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5
    e0.record()
    for _ in range(it):
        ops.gemm(a, w, out, mode=mode, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name:12s} M={M:7d} N={N:6d} K={K:6d} {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TF/s", flush=True)
    del a, w, out
