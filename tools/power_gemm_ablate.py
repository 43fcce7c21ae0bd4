#!/usr/bin/env python3
"""Package power / shader clock / rate of ONE GEMM shape in a loop, for the ablation switches of the profiling library:
    VITG_PROFILING_LIB=1 VITG_PROFILING=1 VITG_GEMM_FLAGS=<bits> python tools/power_gemm_ablate.py [M N K mode]
bits: 0 shipped, 2 no output stores, 64 operand stream off after the first K-tile (MFMA + LDS fragment reads only), 66 both.
Results are wrong by construction with any bit set; the point is where the 1400 W go (profiles/r02_gemm_timeline.md)."""
import os, sys, time
os.environ.setdefault("VITG_PROFILING_LIB", "1"); os.environ.setdefault("VITG_PROFILING", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoitg_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from power_trace import Sampler, find_hwmon        # noqa: E402  (its module-level demo is guarded below)

M, N, K, mode = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (8192, 8192, 8192, 0)
dev = torch.device("cuda:0")
a = torch.randn((M, K), device=dev).half()
w = (torch.randn((N, K), device=dev) * K ** -0.5).half()
o = torch.zeros((M, N // 2 if mode == 2 else N), dtype=torch.float32 if mode == 1 else torch.float16, device=dev)
res = o if mode == 1 else None
hw = find_hwmon()
if not hw or not hw.get("matched"): print("WARNING: hwmon not matched to cuda:0 by PCI address:", hw, flush=True)
for _ in range(3): ops.gemm(a, w, o, mode=mode, res=res)
torch.cuda.synchronize()
s = Sampler(hw); s.start()
t0, n = time.perf_counter(), 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(20): ops.gemm(a, w, o, mode=mode, res=res)
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
s.stop_ = True; s.join()
p = s.p[len(s.p) // 4:]; f = s.f[len(s.f) // 4:]
tf = 2.0 * M * N * K * n / dt / 1e12
busy = tf * 1e12 / (256 * 4 * 1024 * (sum(f) / len(f)) * 1e9) if f else float("nan")
print(f"flags {os.environ.get('VITG_GEMM_FLAGS', '0'):>3s}  {M}x{N}x{K} mode {mode}: {tf:7.1f} TF/s  power {sum(p) / len(p):7.1f} W  sclk {sum(f) / len(f):.3f} GHz  "
      f"-> matrix-pipe occupancy {100 * busy:.0f} %", flush=True)
