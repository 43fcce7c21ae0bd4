#!/bin/bash
# round 4, GPU call 13: tile-order group size of the GEMMs re-measured on this round's kernels (profiling library, both arms), in-step
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c13"; mkdir -p "$OUT"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:10s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  GEMM {d['roofline']['achieved']} TF/s  parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
}
for r in 1 2 3; do for g in 4 6 8 3; do one "g${g}_$r" VITG_PROFILING_LIB=1 VITG_GEMM_GROUP=$g; done; done | tee "$OUT/step_ab.txt"
