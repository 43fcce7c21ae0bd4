#!/usr/bin/env python3
"""GPU: the headline workload through the operand-split tier, `n` forwards (argv[1], default 2) after one warm-up - for timing and for
`rocprofv3 --kernel-trace --stats` (tools/gpu_call.sh ... "py:r06/split_step.py 2")."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoitg_amd import ops, synth
from videoitg_amd.config import ITGConfig
from videoitg_amd.engine import SelectorEngine
from videoitg_amd.synth_device import packed_random_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = ITGConfig()
dev = torch.device("cuda:0")
eng = SelectorEngine(packed_random_weights(cfg, torch.float16, dev, seed=0, vit_q_prescaled=False), operand_split=True)
ids, pix = synth.workload_inputs(cfg, 512, 448, 60)
pix = pix.half().to(dev)
def step():
    lg = eng.forward(ids, None, [pix])[0].view(-1)
    return ops.rank(lg)[1].cpu().tolist()
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"operand-split tier: {dt * 1e3:.1f} ms per 512 x 448^2 video = {512 / dt:.1f} frames/s")
