#!/usr/bin/env python3
"""hd-128 decoder attention (all 12 860 keys, GQA 28:4) with FEWER and fewer 256-query blocks, i.e. busy CUs: package power, shader clock and rate per
busy CU (hwmon of cuda:0, tools/power_trace.py).  Behind profiles/r04_experiments.md section 11: are the idle CUs of one context-parallel rank's launch
(196 blocks on 256 CUs) lost time, or do the busy ones run faster inside the same package power?"""
import os, sys, time, torch
os.environ.setdefault("VITG_PROFILING_LIB", "1")   # attn_waves = 8 pins the 256-query blocks (a research knob: profiling library)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from videoitg_amd import ops
from power_trace import Sampler, find_hwmon

ops.set_option("attn_waves", 8)
hw = find_hwmon()
print("hwmon matched to cuda:0:", bool(hw and hw.get("matched")), flush=True)
dt = torch.float16
H, HKV, D, Sk = 28, 4, 128, 12860
kv = torch.randn((Sk, 2 * HKV * D), device="cuda").to(dt)
for S in (12860, 2304, 1660, 1024, 512, 256):
    q = torch.randn((S, H * D), device="cuda").to(dt)
    o = torch.empty((S, H * D), dtype=dt, device="cuda")
    f = lambda: ops.attention(q, kv, kv[:, HKV * D:], o, batch=1, heads=H, kv_heads=HKV, seq=S, kv_seq=Sk, head_dim=D, q_strides=(0, H * D, D),
                              k_strides=(0, 2 * HKV * D, D), v_strides=(0, 2 * HKV * D, D), o_strides=(0, H * D, D), scale=D ** -0.5)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = Sampler(hw); s.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(20): f()
        torch.cuda.synchronize(); n += 20
    dtm = (time.perf_counter() - t0) / n
    s.stop_ = True; s.join()
    p = s.p[len(s.p) // 4:]; fq = s.f[len(s.f) // 4:]
    blocks = -(-S // 256) * H
    busy = min(blocks, 256)
    tf = 4.0 * S * Sk * D * H / dtm / 1e12
    print(f"S = {S:5d}: {blocks:4d} blocks of 256 queries ({busy:3d} CUs busy{'' if blocks <= 256 else ', %.2f rounds' % (blocks / 256)}): {dtm * 1e6:8.1f} us  {tf:7.1f} TF/s  "
          f"{tf / busy * (1.0 if blocks <= 256 else -(-blocks // 256) * 256 / blocks):5.2f} TF/s per busy CU   power {sum(p) / len(p):7.1f} W  sclk {sum(fq) / len(fq):.3f} GHz", flush=True)
