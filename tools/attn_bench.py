#!/usr/bin/env python3
"""Attention kernel throughput on the production shapes (random data).
python tools/attn_bench.py [fp16|bf16] [opt=value ...]     e.g.  attn_q64=1 attn_waves=8"""
import os
import sys
import torch

if any("=" in a and not a.startswith("variant=") for a in sys.argv[1:]):       # option knobs other than the product's four live in the profiling library
    os.environ.setdefault("VITG_PROFILING_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops  # noqa: E402

dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float16
variant = 0
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        if k == "variant":
            variant = int(v)             # 2 = q pre-multiplied by scale * log2(e) (vitg_attention variant bit 1)
        else:
            ops.set_option(k, int(v))
for name, B, H, HKV, S, D in [("vit448", 512, 16, 16, 1024, 72), ("vit384", 512, 16, 16, 729, 72), ("vit448/8", 64, 16, 16, 1024, 72),
                              ("llm", 1, 28, 4, 12860, 128)]:
    ld = (H + 2 * HKV) * D
    qkv = torch.randn((B * S, ld), device="cuda").to(dt)
    o = torch.empty((B * S, H * D), dtype=dt, device="cuda")
    args = dict(batch=B, heads=H, kv_heads=HKV, seq=S, head_dim=D, q_strides=(S * ld, ld, D), k_strides=(S * ld, ld, D),
                v_strides=(S * ld, ld, D), o_strides=(S * H * D, H * D, D), scale=D ** -0.5, variant=variant)
    f = lambda: ops.attention(qkv, qkv[:, H * D:], qkv[:, (H + HKV) * D:], o, **args)
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:8s} B={B} H={H}/{HKV} S={S} D={D}: {ms:7.3f} ms  {4.0 * B * H * S * S * D / ms / 1e9:7.1f} TF/s (algorithmic)", flush=True)
