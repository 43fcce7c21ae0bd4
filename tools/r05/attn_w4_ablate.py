#!/usr/bin/env python3
"""Timing ablations of the one-wave-per-SIMD hd-128 kernel (attention_w4.hip) on the LLM shape: which stream bounds it?
attn_sp = 100 + bits (profiling library): 1 no exponent stream, 2 no K/V staging, 4 no MFMAs, 8 no fragment reads, 16 no barrier.  Results are WRONG."""
import os, sys, torch
os.environ["VITG_PROFILING_LIB"] = "1"
os.environ["VITG_PROFILING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops
B, H, HKV, S, D = 1, 28, 4, 12860, 128
ld = (H + 2 * HKV) * D
qkv = torch.randn((B * S, ld), device="cuda").half()
o = torch.empty((B * S, H * D), dtype=torch.float16, device="cuda")
def run(sp):
    ops.set_option("attn_sp", sp)
    ops.attention(qkv, qkv[:, H * D:], qkv[:, (H + HKV) * D:], o, batch=B, heads=H, kv_heads=HKV, seq=S, head_dim=D,
                  q_strides=(S * ld, ld, D), k_strides=(S * ld, ld, D), v_strides=(S * ld, ld, D), o_strides=(S * H * D, H * D, D),
                  scale=D ** -0.5, variant=0)
names = {0: "shipped sp kernel", 4: "w4 PF2", 5: "w4 PF3", 101: "w4 - exponents", 102: "w4 - staging", 104: "w4 - MFMAs", 108: "w4 - fragment reads",
         116: "w4 - barrier", 118: "w4 - staging - barrier", 105: "w4 - exponents - MFMAs", 119: "w4 - exponents - staging - barrier"}
for rnd in range(2):
    for sp, nm in names.items():
        for _ in range(2): run(sp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(sp)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"round {rnd}  {nm:36s} {ms:6.3f} ms  {4.0 * B * H * S * S * D / ms / 1e9:7.1f} TF/s-equivalent", flush=True)
