#!/usr/bin/env python3
"""Where a 256-tile GEMM block spends its time: in-kernel timeline of the shipped kernel (profiling library only).

    VITG_PROFILING_LIB=1 python tools/gemm_timeline.py [fp16|bf16]

Waves 0 and 4 (the two wave groups of the phased schedule) of the first 64 blocks stamp the 100 MHz real-time counter at the phase
boundaries of their first four tiles and after every K-tile (`vitg_gemm_timeline_buffer`, gemm256.hip `stamp`).  Printed per shape, in
microseconds averaged over blocks (tiles 1-2 = steady state of the tile loop): prologue (tile start -> first K-tile landed), main loop,
epilogue (-> stores issued), hand-over (-> next tile start), the K-tile period (mean / p90 / max) and the first K-tiles individually.
"""
import ctypes as C, os, sys
os.environ.setdefault("VITG_PROFILING_LIB", "1")
os.environ.setdefault("VITG_PROFILING", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import _lib, ops

lib = _lib.load()
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
NB = 64
shapes = [("out_32cu", 8192, 1152, 1152, 1, 0), ("out_96cu", 24576, 1152, 1152, 1, 0), ("qkv_90cu", 2560, 3456, 1152, 0, 0), ("vit_qkv", 524288, 3456, 1152, 0, 0), ("vit_fc1", 524288, 4352, 1152, 0, ops.ACT_GELU_TANH), ("vit_out", 524288, 1152, 1152, 1, 0),
          ("vit_fc2", 524288, 1152, 4352, 1, 0), ("llm_qkv", 12860, 4608, 3584, 0, 0), ("llm_gu", 12860, 37888, 3584, 2, 0),
          ("llm_down", 12860, 3584, 18944, 1, 0), ("8192^3", 8192, 8192, 8192, 0, 0)]
if len(sys.argv) > 2: shapes = [s for s in shapes if s[0] in sys.argv[2].split(",")]
for name, M, N, K, mode, act in shapes:
    a = torch.randn((M, K), device="cuda").to(dt)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(dt)
    bias = torch.randn(N, device="cuda") * 0.02
    out = torch.zeros((M, N if mode != 2 else N // 2), dtype=torch.float32 if mode == 1 else dt, device="cuda")
    res = out if mode == 1 else None
    kw = dict(mode=mode, res=res, bias=None if mode == 2 else bias, act=act)
    for _ in range(2): ops.gemm(a, w, out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.gemm(a, w, out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    buf = torch.zeros(NB * 2 * 4 * 32, dtype=torch.int64, device="cuda")
    lib.vitg_gemm_timeline_buffer(C.c_void_p(buf.data_ptr()), NB)
    ops.gemm(a, w, out, **kw)
    torch.cuda.synchronize()
    lib.vitg_gemm_timeline_buffer(None, 0)
    t = buf.cpu().view(NB, 2, 4, 32).double() * 0.01          # microseconds
    nk = K // 64
    print(f"== {name}  M={M} N={N} K={K} mode={mode} act={act}  {2.0 * M * N * K / ms / 1e9:.0f} TF/s ({ms:.3f} ms), {nk} K-tiles per tile", flush=True)
    for grp in (0, 1):
        for tl in (0, 1, 2):
            x = t[:, grp, tl]
            ok = (x[:, 30] > 0) & (x[:, 0] > 0)
            if ok.sum() == 0: continue
            x = x[ok]
            nxt = t[:, grp, tl + 1][ok]
            last = 2 + min(nk, 24) - 1
            pro, main, epi = x[:, 1] - x[:, 0], x[:, 28] - x[:, 1], x[:, 29] - x[:, 28]
            done = x[:, 30] - x[:, 29]
            hand = (nxt[:, 0] - x[:, 30])[nxt[:, 0] > 0]
            kts = torch.cat([(x[:, 2] - x[:, 1])[:, None], x[:, 3:last + 1] - x[:, 2:last]], 1)       # per K-tile durations
            total = x[:, 30] - x[:, 0]
            print(f"  group {'AB'[grp]} tile {tl} ({int(ok.sum())} blocks): total {total.mean():6.2f} us = prologue {pro.mean():5.2f} + main {main.mean():6.2f} "
                  f"+ epilogue {epi.mean():5.2f} + drain {done.mean():5.2f}" + (f"  | hand-over {hand.mean():5.2f}" if len(hand) else "")
                  + f"  | K-tile mean {kts.mean():.3f} p90 {kts.flatten().quantile(0.9):.3f} max {kts.max():.2f}  first four {[round(float(v), 2) for v in kts.mean(0)[:4]]}", flush=True)
    del a, w, out, buf
