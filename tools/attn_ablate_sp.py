import os, sys, torch
os.environ['VITG_PROFILING'] = '1'; os.environ.setdefault('VITG_PROFILING_LIB', '1')          # ablation switches produce wrong results: opt in explicitly
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops
ops.set_option("attn_sp", 1)
dt = torch.float16
for name, B, H, HKV, S, D in [("vit448", 512, 16, 16, 1024, 72), ("llm", 1, 28, 4, 12860, 128)]:
    ld = (H + 2 * HKV) * D
    qkv = torch.randn((B * S, ld), device="cuda").to(dt)
    o = torch.empty((B * S, H * D), dtype=dt, device="cuda")
    for abl, label in [(0, "full"), (8, "no DMA in loop"), (16, "V^T fragments not re-read"), (24, "neither")]:
        args = dict(batch=B, heads=H, kv_heads=HKV, seq=S, head_dim=D, q_strides=(S * ld, ld, D), k_strides=(S * ld, ld, D),
                    v_strides=(S * ld, ld, D), o_strides=(S * H * D, H * D, D), scale=D ** -0.5, variant=abl << 4)
        f = lambda: ops.attention(qkv, qkv[:, H * D:], qkv[:, (H + HKV) * D:], o, **args)
        f(); f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): f()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:7s} {label:28s} {e0.elapsed_time(e1)/3:7.3f} ms", flush=True)
