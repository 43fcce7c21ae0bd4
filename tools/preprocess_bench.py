#!/usr/bin/env python3
"""GPU preprocessing front-end throughput (uint8 NHWC frames -> normalised NCHW fp16), HBM roofline.
    python tools/preprocess_bench.py [T] [H] [W] [S]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops  # noqa: E402

args = [int(a) for a in sys.argv[1:]]
T, H, W, S = (args + [512, 720, 1280, 448][len(args):])[:4]
fr = torch.randint(0, 256, (T, H, W, 3), dtype=torch.uint8, device="cuda")
for dt in (torch.float16, torch.float32):
    for _ in range(2):
        out = ops.preprocess_u8(fr, S, dt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = ops.preprocess_u8(fr, S, dt)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    alg = T * H * W * 3 + out.numel() * out.element_size()
    print(f"T={T} {H}x{W} -> {S}^2 {dt}: {ms:7.3f} ms  {T / ms * 1e3:9.0f} frames/s  algorithmic {alg / 1e9:.2f} GB -> {alg / ms / 1e6:7.1f} GB/s "
          f"({alg / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s)", flush=True)
