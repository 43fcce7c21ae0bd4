#!/usr/bin/env python3
"""Same-call A/B: the shipped 8-wave GEMM (vitg_gemm, 16-bit output, no activation) against the one-wave-per-SIMD probe (tools/ubench/gemm_w4r) on the
same shapes and the same kind of data (A ~ U(-1, 1), W ~ U(-1, 1) / sqrt(K), fp16), alternated.   python tools/r05/w4r_ab.py [rounds]"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from videoitg_amd import ops  # noqa: E402

SHAPES = [("square8k", 8192, 8192, 8192), ("tower_fc1", 524288, 4352, 1152), ("tower_qkv", 524288, 3584, 1152), ("decoder_gateup", 12800, 37888, 3584)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
exe = os.path.join(ROOT, "tools", "ubench", sys.argv[2] if len(sys.argv) > 2 else "gemm_w4r")


def shipped(M, N, K):
    a = (torch.rand((M, K), device="cuda") * 2 - 1).half()
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) * K ** -0.5).half()
    out = torch.empty((M, N), dtype=torch.float16, device="cuda")
    for _ in range(2):
        ops.gemm(a, w, out, mode=ops.GEMM_STORE16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 10
    e0.record()
    for _ in range(it):
        ops.gemm(a, w, out, mode=ops.GEMM_STORE16)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    del a, w, out
    torch.cuda.empty_cache()
    return 2.0 * M * N * K / ms / 1e9


for name, M, N, K in SHAPES:
    for r in range(rounds):
        tf = shipped(M, N, K)
        p = subprocess.run([exe, str(M), str(N), str(K), "10", "1" if r == 0 else "0"], capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if "TFLOP/s" in l]
        chk = [l for l in p.stdout.splitlines() if l.startswith("check")]
        print(f"{name:16s} {M}x{N}x{K} round {r}: shipped {tf:7.1f} TF/s | probe {line[0].split(':')[1].strip() if line else 'FAILED ' + p.stderr[-300:]} {chk[0] if chk else ''}", flush=True)
