#!/usr/bin/env python3
"""Why is the one-wave-per-SIMD hd-128 kernel (attn_sp = 4) faster than the 8-wave one in isolation and slower inside the step?  Time the attention launch
alone (HIP events around it) when OTHER kernels run between its launches: nothing | a light kernel (RMSNorm) | a heavy GEMM (the decoder's gate/up)."""
import os, sys, torch
os.environ.setdefault("VITG_PROFILING_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops
B, H, HKV, S, D = 1, 28, 4, 12860, 128
ld = (H + 2 * HKV) * D
dt = torch.float16
qkv = torch.randn((S, ld), device="cuda").to(dt)
o = torch.empty((S, H * D), dtype=dt, device="cuda")
x32 = torch.randn((S, 3584), device="cuda")
w32 = torch.ones(3584, device="cuda")
hn = torch.empty((S, 3584), dtype=dt, device="cuda")
wg = (torch.randn((37888, 3584), device="cuda") * 0.02).to(dt)
f = torch.empty((S, 18944), dtype=dt, device="cuda")
def attn(sp):
    ops.set_option("attn_sp", sp)
    ops.attention(qkv, qkv[:, H * D:], qkv[:, (H + HKV) * D:], o, batch=B, heads=H, kv_heads=HKV, seq=S, head_dim=D, q_strides=(S * ld, ld, D),
                  k_strides=(S * ld, ld, D), v_strides=(S * ld, ld, D), o_strides=(S * H * D, H * D, D), scale=D ** -0.5, variant=0)
def light(): ops.rmsnorm(x32, w32, hn, 3584, 1e-6)
def heavy(): ops.gemm(hn, wg, f, mode=ops.GEMM_SWIGLU)
light(); heavy()
for rnd in range(3):
    for between, name in ((None, "nothing"), (light, "RMSNorm"), (heavy, "gate/up GEMM"), (lambda: (heavy(), heavy(), heavy()), "3 x gate/up GEMM")):
        line = []
        for sp in (0, 4):
            for _ in range(3):
                if between: between()
                attn(sp)
            tot = 0.0
            for _ in range(10):
                if between: between()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); attn(sp); e1.record(); torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            line.append(f"attn_sp={sp}: {tot / 10:6.3f} ms")
        print(f"round {rnd}  between launches: {name:18s} " + "   ".join(line), flush=True)
ops.set_option("attn_sp", 0)
