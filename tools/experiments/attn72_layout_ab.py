#!/usr/bin/env python3
"""hd-72 tower attention: K/V read from the packed token-major qkv rows (what the qkv GEMM writes: a key row = 144 B at a stride of 6 912 B, straddling
two 128-B lines) against a head-major layout ([frame, head, token, 72]: a 64-key tile = 9 216 contiguous bytes).  Same kernel, strides only.
python tools/experiments/attn72_layout_ab.py [frames=512]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H, D, P = 16, 72, 1024
dt = torch.float16
c = D ** -0.5 * 1.4426950408889634
qkv = (torch.randn((T * P, 3 * H * D), device="cuda") * 1.0).to(dt)
qkv[:, :H * D] *= c
o = torch.zeros((T * P, 1152), dtype=dt, device="cuda")
# head-major copies: [part][T, H, P, D]
hm = qkv.view(T, P, 3, H, D).permute(2, 0, 3, 1, 4).contiguous()
qh, kh, vh = hm[0], hm[1], hm[2]
o2 = torch.zeros((T * P, 1152), dtype=dt, device="cuda")
ld = 3 * H * D
def packed():
    ops.attention(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, batch=T, heads=H, kv_heads=H, seq=P, head_dim=D, q_strides=(P * ld, ld, D),
                  k_strides=(P * ld, ld, D), v_strides=(P * ld, ld, D), o_strides=(P * 1152, 1152, D), scale=D ** -0.5, variant=2)
def kv_headmajor():
    ops.attention(qkv, kh, vh, o2, batch=T, heads=H, kv_heads=H, seq=P, head_dim=D, q_strides=(P * ld, ld, D),
                  k_strides=(H * P * D, D, P * D), v_strides=(H * P * D, D, P * D), o_strides=(P * 1152, 1152, D), scale=D ** -0.5, variant=2)
def all_headmajor():
    ops.attention(qh, kh, vh, o2, batch=T, heads=H, kv_heads=H, seq=P, head_dim=D, q_strides=(H * P * D, D, P * D),
                  k_strides=(H * P * D, D, P * D), v_strides=(H * P * D, D, P * D), o_strides=(P * 1152, 1152, D), scale=D ** -0.5, variant=2)
def t(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
fl = 4.0 * T * H * P * P * D
for r in range(3):
    a, b, cc = t(packed), t(kv_headmajor), t(all_headmajor)
    print(f"round {r}: packed {a:.3f} ms ({fl / a / 1e9:.0f} TF/s)   K|V head-major {b:.3f} ms ({100 * (a / b - 1):+.1f} %)   Q|K|V head-major {cc:.3f} ms ({100 * (a / cc - 1):+.1f} %)", flush=True)
print("max |diff| packed vs head-major:", (o.float() - o2.float()).abs().max().item())
