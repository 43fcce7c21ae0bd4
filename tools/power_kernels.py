#!/usr/bin/env python3
"""Package power and shader clock under each kernel class of the step, timed alone in a loop (hwmon of cuda:0, tools/power_trace.py):
the two attention kernels, LayerNorm, and the GEMM shapes.  Shows which kernels sit on the 1400 W cap (energy-bound: a schedule gain
returns as a lower clock) and which do not (latency / bandwidth-bound: a schedule gain is a time gain)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from videoitg_amd import ops
from power_trace import Sampler, find_hwmon

hw = find_hwmon()
print("hwmon matched to cuda:0:", bool(hw and hw.get("matched")), flush=True)
dt = torch.float16


def run(name, f, work, unit, secs=4.0):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s = Sampler(hw); s.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        for _ in range(10): f()
        torch.cuda.synchronize(); n += 10
    dtm = time.perf_counter() - t0
    s.stop_ = True; s.join()
    p = s.p[len(s.p) // 4:]; fq = s.f[len(s.f) // 4:]
    print(f"{name:44s}: {dtm / n * 1e3:7.3f} ms  {work / (dtm / n) / 1e12:8.1f} {unit}  power {sum(p) / len(p):7.1f} W  sclk {sum(fq) / len(fq):.3f} GHz", flush=True)


for name, B, H, HKV, S, D in [("attention hd72 (tower, 512 x 1024 tokens)", 512, 16, 16, 1024, 72), ("attention hd128 (LLM, 12860 tokens, GQA 28:4)", 1, 28, 4, 12860, 128)]:
    ld = (H + 2 * HKV) * D
    qkv = torch.randn((B * S, ld), device="cuda").to(dt)
    o = torch.empty((B * S, H * D), dtype=dt, device="cuda")
    args = dict(batch=B, heads=H, kv_heads=HKV, seq=S, head_dim=D, q_strides=(S * ld, ld, D), k_strides=(S * ld, ld, D),
                v_strides=(S * ld, ld, D), o_strides=(S * H * D, H * D, D), scale=D ** -0.5)
    run(name, lambda: ops.attention(qkv, qkv[:, H * D:], qkv[:, (H + HKV) * D:], o, **args), 4.0 * B * H * S * S * D, "TF/s")
    del qkv, o
x = torch.randn((524288, 1152), device="cuda"); w = torch.ones(1152, device="cuda"); b = torch.zeros(1152, device="cuda")
y = torch.empty((524288, 1152), dtype=dt, device="cuda")
run("layernorm 524288 x 1152 (fp32 -> fp16)", lambda: ops.layernorm(x, w, b, y, 1152, 1e-6), 524288 * 1152 * 6.0, "TB/s")
del x, y
for name, M, N, K, mode in [("GEMM vit_qkv", 524288, 3456, 1152, 0), ("GEMM vit_out (fp32 residual)", 524288, 1152, 1152, 1), ("GEMM vit_fc2 (fp32 residual)", 524288, 1152, 4352, 1),
                            ("GEMM llm gate/up (SwiGLU)", 12860, 37888, 3584, 2), ("GEMM llm down (fp32 residual)", 12860, 3584, 18944, 1)]:
    a = torch.randn((M, K), device="cuda").to(dt); wt = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    out = torch.zeros((M, N // 2 if mode == 2 else N), dtype=torch.float32 if mode == 1 else dt, device="cuda")
    run(name, lambda: ops.gemm(a, wt, out, mode=mode, res=out if mode == 1 else None), 2.0 * M * N * K, "TF/s")
    del a, wt, out
