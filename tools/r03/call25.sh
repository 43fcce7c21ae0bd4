#!/bin/bash
# round 3, GPU call 25: bias test hoisted out of the quad loop of the interior 16-bit stages; gemm_flags bit30 = per quad
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c25"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -k "gemm" -x -q > "$OUT/pytest_gemm.txt" 2>&1; tail -3 "$OUT/pytest_gemm.txt"
for r in 1 2 3; do
  for fl in 0 1073741824; do
    VITG_GEMM_FLAGS=$fl timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_fl${fl}_$r.json" 2> "$OUT/bench_fl${fl}_$r.err"
    python - "$OUT/bench_fl${fl}_$r.json" $fl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"gemm_flags={sys.argv[2]}: {d['value']} frames/s {d['ms_per_step']} ms  GEMM {d['roofline']['achieved']} TF/s frac {d['roofline']['frac']} parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
  done
done | tee "$OUT/step_ab.txt"
