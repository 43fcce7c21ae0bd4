#!/usr/bin/env python3
"""hd-128 attention on the shapes ONE rank of the context-parallel LLM sees (S local queries x all 12860 keys, GQA 28:4) for the block
sizes attn_waves = 3 / 4 / 8 (96 / 128 / 256 queries per block): which one balances 256 CUs best at each rank count."""
import os, sys, torch
os.environ.setdefault("VITG_PROFILING_LIB", "1")   # research knobs (round 4: the product library takes four option keys only)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops
dt = torch.float16
H, HKV, D, Sk = 28, 4, 128, 12860
kv = torch.randn((Sk, 2 * HKV * D), device="cuda").to(dt)
for world, S in [(8, 1660), (8, 1600), (4, 3260), (2, 6460), (1, 12860)]:
    q = torch.randn((S, H * D), device="cuda").to(dt)
    o = torch.empty((S, H * D), dtype=dt, device="cuda")
    row = []
    for nw in (0, 3, 4, 8):
        ops.set_option("attn_waves", nw)
        f = lambda: ops.attention(q, kv, kv[:, HKV * D:], o, batch=1, heads=H, kv_heads=HKV, seq=S, kv_seq=Sk, head_dim=D,
                                  q_strides=(0, H * D, D), k_strides=(0, 2 * HKV * D, D), v_strides=(0, 2 * HKV * D, D), o_strides=(0, H * D, D), scale=D ** -0.5)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 10)
    ops.set_option("attn_waves", 0)
    print(f"world {world}: S = {S:5d} queries: auto {row[0]*1e3:7.1f} us   3 waves {row[1]*1e3:7.1f}   4 waves {row[2]*1e3:7.1f}   8 waves {row[3]*1e3:7.1f}   "
          f"(ideal 1/{world} of the single-GPU launch: {2468.0 / world * S / (12860 / world):7.1f} us)", flush=True)
