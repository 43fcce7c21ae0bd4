#!/usr/bin/env python3
"""Is the one-wave-per-SIMD hd-128 kernel (attn_sp = 4) bit-identical to the 8-wave software-pipelined one (attn_sp = 0)?  Per key count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_w4_ab import run
for dt in (torch.float16, torch.bfloat16):
    for S in (1, 63, 64, 65, 128, 192, 256, 300, 1000, 2049, 12860):
        B, H, HKV, D = 1, 4, 2, 128
        torch.manual_seed(S)
        qkv = torch.randn((B * S, (H + 2 * HKV) * D), device="cuda").to(dt)
        outs = []
        for sp in (0, 4):
            o = torch.zeros((B * S, H * D), dtype=dt, device="cuda")
            run(qkv, o, B, H, HKV, S, D, sp)
            outs.append(o.float())
        d = (outs[0] - outs[1]).abs()
        print(f"{str(dt):15s} S={S:6d}: max|w4 - sp| = {d.max().item():.3e}   differing elements {int((d > 0).sum())} of {d.numel()}", flush=True)
# where do they differ?  (fp16)
from attn_w4_ab import ref
for S in (63, 64, 300, 2049):
    B, H, HKV, D = 1, 4, 2, 128
    torch.manual_seed(S)
    qkv = torch.randn((B * S, (H + 2 * HKV) * D), device="cuda").half()
    outs = []
    for sp in (0, 4):
        o = torch.zeros((B * S, H * D), dtype=torch.float16, device="cuda")
        run(qkv, o, B, H, HKV, S, D, sp)
        outs.append(o.float().view(S, H, D))
    r = ref(qkv, B, H, HKV, S, D).view(S, H, D)
    idx = (outs[0] != outs[1]).nonzero()
    for q, h, d in idx[:12].tolist():
        print(f"S={S} q={q} (q%64={q % 64}) head={h} d={d}: sp {outs[0][q, h, d].item():.6f} w4 {outs[1][q, h, d].item():.6f} ref {r[q, h, d].item():.7f}")
