#!/bin/bash
# round 3, GPU call 8: per-rank (small-M) GEMM shapes of the context-parallel LLM on the 16x16x32 kernels; fc1 on the tall tile in the step
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c8"; mkdir -p "$OUT"
timeout 900 python tools/gemm_small_m.py fp16 2>&1 | grep -v amdgpu.ids | tee "$OUT/gemm_small_m.txt"
cat > /tmp/q.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from videoitg_amd import ops
dt = torch.float16
for M in (1660, 3260, 6460):
    for name, N, K in [("q", 3584, 3584), ("kv", 1024, 3584)]:
        a = (torch.rand((M, K), device="cuda") * 2 - 1).to(dt); w = ((torch.rand((N, K), device="cuda") * 2 - 1) * K ** -0.5).to(dt)
        out = torch.empty((M, N), dtype=dt, device="cuda"); res = []
        for tile, cfg in [(0, 0), (128, 0), (256, 6), (256, 2), (256, 9), (256, 10)]:
            ops.set_option("gemm_tile", tile); ops.set_option("gemm_cfg", cfg)
            for _ in range(3): ops.gemm(a, w, out, mode=0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm(a, w, out, mode=0)
            e1.record(); torch.cuda.synchronize()
            res.append(f"{e0.elapsed_time(e1) / 10 * 1e3:6.1f}")
        ops.set_option("gemm_tile", 0); ops.set_option("gemm_cfg", 0)
        print(f"M={M:5d} {name:3s} us: auto {res[0]} | tile128 {res[1]} | cfg6 256x256 {res[2]} | cfg2 256x128 {res[3]} | cfg9 256x384 {res[4]} | cfg10 384x256 {res[5]}", flush=True)
PY
timeout 600 python /tmp/q.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/gemm_q_kv.txt"
for r in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_r$r.json" 2> "$OUT/bench_r$r.err"
  python - <<PY
import json
r = json.loads(open("$OUT/bench_r$r.json").read().strip().splitlines()[-1])
o = r.get("roofline_other", {})
print("default run $r:", r["value"], "frames/s", r["ms_per_step"], "ms  GEMM", r["roofline"]["achieved"], "TF/s frac", r["roofline"]["frac"], "parity", r["parity"]["max_abs_dscore"], r["parity"]["pass"], "patchify", o.get("patchify", {}).get("achieved"), flush=True)
PY
done
