#!/usr/bin/env python3
"""Is the hd-128 A/B data dependent?  Same shape, different score statistics (the lazy rescale fires more often on peaky scores; zeros draw less power)."""
import os, sys, torch
os.environ.setdefault("VITG_PROFILING_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops
B, H, HKV, S, D = 1, 28, 4, 12860, 128
ld = (H + 2 * HKV) * D
dt = torch.float16
o = torch.empty((S, H * D), dtype=dt, device="cuda")
def attn(qkv, sp):
    ops.set_option("attn_sp", sp)
    ops.attention(qkv, qkv[:, H * D:], qkv[:, (H + HKV) * D:], o, batch=B, heads=H, kv_heads=HKV, seq=S, head_dim=D, q_strides=(S * ld, ld, D),
                  k_strides=(S * ld, ld, D), v_strides=(S * ld, ld, D), o_strides=(S * H * D, H * D, D), scale=D ** -0.5, variant=0)
torch.manual_seed(0)
base = torch.randn((S, ld), device="cuda")
ramp = torch.linspace(0, 1, S, device="cuda")[:, None]
variants = {"N(0,1)": base, "Q x 4 (peaky)": torch.cat([base[:, :H * D] * 4, base[:, H * D:]], 1), "Q x 16": torch.cat([base[:, :H * D] * 16, base[:, H * D:]], 1),
            "Q x 0.05 (flat)": torch.cat([base[:, :H * D] * 0.05, base[:, H * D:]], 1),
            "K grows along the sequence (max keeps moving)": torch.cat([base[:, :H * D].abs(), base[:, H * D:(H + HKV) * D].abs() * (0.2 + 8 * ramp), base[:, (H + HKV) * D:]], 1),
            "zeros": base * 0}
for name, x in variants.items():
    qkv = x.to(dt)
    line = []
    for sp in (0, 4, 0, 4):
        for _ in range(2): attn(qkv, sp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): attn(qkv, sp)
        e1.record(); torch.cuda.synchronize()
        line.append(f"sp={sp}: {e0.elapsed_time(e1) / 10:6.3f}")
    print(f"{name:48s} " + "  ".join(line), flush=True)
ops.set_option("attn_sp", 0)
