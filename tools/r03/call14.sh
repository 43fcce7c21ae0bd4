#!/bin/bash
# round 3, GPU call 14: 384x256 tile as a 4 x 2 wave grid (A/B per shape, in the step), refreshed vendor comparison
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c14"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "wm4 or wm2" -x -q > "$OUT/pytest_wm4.txt" 2>&1; tail -3 "$OUT/pytest_wm4.txt"
timeout 900 python tools/gemm_wm4_ab.py fp16 3 2>&1 | grep -v amdgpu.ids | tee "$OUT/gemm_wm4_ab.txt"
for r in 1 2; do
  for wm in -1 1 4 5 7 15; do
    VITG_GEMM_WM4=$wm timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_wm${wm}_$r.json" 2> "$OUT/bench_wm${wm}_$r.err"
    python - "$OUT/bench_wm${wm}_$r.json" $wm <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"wm4={sys.argv[2]}: {d['value']} frames/s {d['ms_per_step']} ms  GEMM {d['roofline']['achieved']} TF/s frac {d['roofline']['frac']} parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
  done
done | tee "$OUT/step_ab.txt"
timeout 900 python tools/gemm_vs_vendor.py fp16 2>&1 | grep -v amdgpu.ids | tee "$OUT/gemm_vs_vendor_fp16.txt"
