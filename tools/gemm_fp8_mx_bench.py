#!/usr/bin/env python3
"""The tower MLP on the fp8 path, per layer: [fc1 + GELU -> fp16] + [per-token quantiser] + [fc2] against the MX pair
[fc1 + GELU -> e4m3 + block scales] + [fc2 on block-scaled activations] (vitg_gemm_fp8_mx), same call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops

def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

M, D, Fv = 524288, 1152, 4352
dt = torch.float16
x = torch.randn((M, D), device="cuda").to(dt)
w1 = (torch.randn((Fv, D), device="cuda") * D ** -0.5).to(dt); w2 = (torch.randn((D, Fv), device="cuda") * Fv ** -0.5).to(dt)
b1 = torch.randn(Fv, device="cuda") * 0.02; b2 = torch.randn(D, device="cuda") * 0.02
x8, sx = ops.quant_rows_fp8(x, D); w18, sw1 = ops.quant_rows_fp8(w1, D); w28, sw2 = ops.quant_rows_fp8(w2, Fv)
f16 = torch.empty((M, Fv), dtype=dt, device="cuda"); f8 = torch.empty((M, Fv), dtype=torch.uint8, device="cuda")
sf = torch.empty(M, dtype=torch.float32, device="cuda"); fsc = torch.empty((Fv // 128, M, 4), dtype=torch.uint8, device="cuda")
res = torch.zeros((M, D), dtype=torch.float32, device="cuda")
t = {}
t["fc1 -> fp16"] = timeit(lambda: ops.gemm_fp8(x8, sx, w18, sw1, f16, mode=0, out_dtype=dt, bias=b1, act=ops.ACT_GELU_TANH))
t["quantiser"] = timeit(lambda: ops.quant_rows_fp8(f16, Fv, f8, sf))
t["fc2 (per-token)"] = timeit(lambda: ops.gemm_fp8(f8, sf, w28, sw2, res, mode=1, out_dtype=dt, bias=b2, res=res))
t["fc1 -> MX e4m3"] = timeit(lambda: ops.gemm_fp8_mx(x8, w18, sw1, f8, out_dtype=dt, sa=sx, o_mx=fsc, bias=b1, act=ops.ACT_GELU_TANH))
t["fc2 (MX)"] = timeit(lambda: ops.gemm_fp8_mx(f8, w28, sw2, res, out_dtype=dt, a_mx=fsc, bias=b2, res=res))
for k, v in t.items(): print(f"{k:18s} {v:7.3f} ms", flush=True)
print(f"per-token chain {t['fc1 -> fp16'] + t['quantiser'] + t['fc2 (per-token)']:.3f} ms   MX chain {t['fc1 -> MX e4m3'] + t['fc2 (MX)']:.3f} ms")

# ---- the decoder MLP (M = 12860 tokens): gate/up + SwiGLU -> down
del x, w1, w2, x8, w18, w28, f16, f8, sf, fsc, res
torch.cuda.empty_cache()
M, D, Fl = 12860, 3584, 18944
x = torch.randn((M, D), device="cuda").to(dt)
wg = (torch.randn((2 * Fl, D), device="cuda") * D ** -0.5).to(dt); wd = (torch.randn((D, Fl), device="cuda") * Fl ** -0.5).to(dt)
x8, sx = ops.quant_rows_fp8(x, D); wg8, swg = ops.quant_rows_fp8(wg, D); wd8, swd = ops.quant_rows_fp8(wd, Fl)
f16 = torch.empty((M, Fl), dtype=dt, device="cuda"); f8 = torch.empty((M, Fl), dtype=torch.uint8, device="cuda")
sf = torch.empty(M, dtype=torch.float32, device="cuda"); fsc = torch.empty((Fl // 128, M, 4), dtype=torch.uint8, device="cuda")
res = torch.zeros((M, D), dtype=torch.float32, device="cuda")
t = {}
t["gate/up -> fp16"] = timeit(lambda: ops.gemm_fp8(x8, sx, wg8, swg, f16, mode=2, out_dtype=dt))
t["quantiser"] = timeit(lambda: ops.quant_rows_fp8(f16, Fl, f8, sf))
t["down (per-token)"] = timeit(lambda: ops.gemm_fp8(f8, sf, wd8, swd, res, mode=1, out_dtype=dt, res=res))
t["gate/up -> MX e4m3"] = timeit(lambda: ops.gemm_fp8_mx(x8, wg8, swg, f8, out_dtype=dt, sa=sx, o_mx=fsc, swiglu=True))
t["down (MX)"] = timeit(lambda: ops.gemm_fp8_mx(f8, wd8, swd, res, out_dtype=dt, a_mx=fsc, res=res))
for k, v in t.items(): print(f"{k:20s} {v:7.3f} ms", flush=True)
print(f"per-token chain {t['gate/up -> fp16'] + t['quantiser'] + t['down (per-token)']:.3f} ms   MX chain {t['gate/up -> MX e4m3'] + t['down (MX)']:.3f} ms")
