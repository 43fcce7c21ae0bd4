#!/usr/bin/env python3
"""The tower's out-proj (16-bit residual form, 524 288 x 1152 x 1152) and fc2 (K = 4352) alone in a loop: time per launch.  Run once per library
(product / VITG_LIB_PATH=...exp_seqtn.so) inside one call.  python tools/experiments/res16_seq_tn_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops
dt = torch.float16
M = 524288
for name, N, K in [("out-proj", 1152, 1152), ("fc2", 1152, 4352)]:
    a = torch.randn((M, K), device="cuda").to(dt); w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    x = torch.randn((M, N), device="cuda").to(dt); b = torch.randn((N,), device="cuda")
    x0 = x.clone()
    ops.gemm_res16(a, w, x, bias=b, res16=x)
    chk = x.float().abs().sum().item()
    for _ in range(3): ops.gemm_res16(a, w, x, bias=b, res16=x)
    ts = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm_res16(a, w, x, bias=b, res16=x0)     # (x0 is never written: every launch reads the same residual)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    print(f"lib={os.path.basename(os.environ.get('VITG_LIB_PATH', 'product'))} {name}: {min(ts):.4f} ms ({2.0 * M * N * K / min(ts) / 1e9:.0f} TF/s)  rounds {[round(t, 4) for t in ts]}  checksum {chk:.6e}", flush=True)
