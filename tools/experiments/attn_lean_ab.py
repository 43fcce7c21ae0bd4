#!/usr/bin/env python3
"""hd-128 decoder attention: the classic launch (one block per q-block) against the launch cut by WORK (vitg_attention_ws, csrc/attention_sp.hip LEAN)
on the single-GPU shape (12 860 queries x 28 heads: 1 428 blocks = 5.58 rounds on 256 CUs) and on what ONE rank of the context-parallel decoder
sees (S local queries x all 12 860 keys).  Alternating, product library.  python tools/attn_lean_ab.py [rounds=3]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dt = torch.float16
H, HKV, D, Sk = 28, 4, 128, 12860
kv = torch.randn((Sk, 2 * HKV * D), device="cuda").to(dt)
ws = torch.empty((ops.attention_ws_bytes(),), dtype=torch.uint8, device="cuda")
print(f"workspace {ws.numel() / 2**20:.1f} MiB")
for world, S in [(1, 12860), (2, 6460), (4, 3260), (8, 1660), (8, 1600)]:
    q = torch.randn((S, H * D), device="cuda").to(dt)
    o = [torch.empty((S, H * D), dtype=dt, device="cuda") for _ in range(2)]
    def run(i, w):
        ops.attention(q, kv, kv[:, HKV * D:], o[i], batch=1, heads=H, kv_heads=HKV, seq=S, kv_seq=Sk, head_dim=D, q_strides=(0, H * D, D),
                      k_strides=(0, 2 * HKV * D, D), v_strides=(0, 2 * HKV * D, D), o_strides=(0, H * D, D), scale=D ** -0.5, ws=w)
    t = [[], []]
    for r in range(rounds):
        for i, w in enumerate((None, ws)):
            for _ in range(3): run(i, w)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(i, w)
            e1.record(); torch.cuda.synchronize()
            t[i].append(e0.elapsed_time(e1) * 100)
    d = (o[0].float() - o[1].float()).abs().max().item()
    flops = 4.0 * S * Sk * D * H
    c, l = min(t[0]), min(t[1])
    print(f"world {world}: S = {S:5d}: classic {c:7.1f} us ({flops / c / 1e6:6.1f} TF/s)   by work {l:7.1f} us ({flops / l / 1e6:6.1f} TF/s)   {100 * (c / l - 1):+5.1f} %   "
          f"max |diff| {d:.2e}   all rounds classic {[round(x, 1) for x in t[0]]} by work {[round(x, 1) for x in t[1]]}", flush=True)
