#!/bin/bash
# round 3, GPU call 31: the profiling-library test file (skipped in the product suite) after the shared-epilogue changes
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c31"; mkdir -p "$OUT"
VITG_PROFILING_LIB=1 timeout 1800 python -m pytest tests/test_gpu_profiling_lib.py -q > "$OUT/pytest_prof.txt" 2>&1; tail -3 "$OUT/pytest_prof.txt"
# quick in-step sweeps on the final kernels: tile-order group size, the 4 x 2 wave grid per epilogue mode
for v in "VITG_GEMM_GROUP=2" "VITG_GEMM_GROUP=4" "VITG_GEMM_GROUP=8" "VITG_GEMM_WM4=1" "VITG_GEMM_WM4=4" "VITG_GEMM_WM4=2" "VITG_GEMM_GROUP=4"; do
  env $v timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_sweep.json" 2> "$OUT/bench_sweep.err"
  python - "$OUT/bench_sweep.json" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]}: {d['value']} frames/s {d['ms_per_step']} ms  GEMM {d['roofline']['achieved']} TF/s")
PY
done | tee "$OUT/sweep.txt"
