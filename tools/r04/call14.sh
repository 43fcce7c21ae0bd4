#!/bin/bash
# round 4, GPU call 14: the residual GEMMs of the tower (out-proj, fc2) with only HALF / NONE of the 16-bit residual tile preloaded into the accumulators
# (the rest added in the epilogue, requested one row pair ahead): does a smaller up-front HBM burst pay?  experiment builds, in-step A/B + residual-GEMM tests
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c14"; mkdir -p "$OUT"
for n in 4 0; do VITG_LIB_PATH="$PWD/videoitg_amd/libvideoitg_hip_exp_pre$n.so" timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "res16" 2>&1 | tail -1; done
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:10s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  GEMM {d['roofline']['achieved']} TF/s  parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
}
for r in 1 2 3; do
  one "all_$r" X=1
  one "half_$r" VITG_LIB_PATH="$PWD/videoitg_amd/libvideoitg_hip_exp_pre4.so"
  one "none_$r" VITG_LIB_PATH="$PWD/videoitg_amd/libvideoitg_hip_exp_pre0.so"
done | tee "$OUT/step_ab.txt"
