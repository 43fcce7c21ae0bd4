#!/bin/bash
# round 4, GPU call 10: fp8 GEMM with the tile loop (one block per CU walking the tile list) against one hardware block per tile (experiment build):
# per shape, in the fp8 job, and the fp8 tests on the new kernel
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c10"; mkdir -p "$OUT"
EXP="$PWD/videoitg_amd/libvideoitg_hip_exp_noloop.so"
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu > "$OUT/pytest_fp8.txt" 2>&1; tail -2 "$OUT/pytest_fp8.txt"
for r in 1 2; do
  echo "## round $r: tile loop (product)"; timeout 300 python tools/gemm_fp8_bench.py 2>&1 | grep -E "M=" | sed -E 's/quant_rows.*//'
  echo "## round $r: one block per tile (round 3 form)"; VITG_LIB_PATH="$EXP" timeout 300 python tools/gemm_fp8_bench.py 2>&1 | grep -E "M=" | sed -E 's/quant_rows.*//'
done | tee "$OUT/gemm_fp8_ab.txt"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --fp8 --fp8-policy all --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:10s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  fp8 GEMM {d['roofline']['achieved']} TF/s = {d['roofline']['frac']}  parity {d['parity']['max_abs_dscore']}")
PY
}
for r in 1 2 3; do one "loop_$r" X=1; one "noloop_$r" VITG_LIB_PATH="$EXP"; done | tee "$OUT/step_ab.txt"
