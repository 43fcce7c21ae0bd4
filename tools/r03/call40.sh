#!/bin/bash
# round 3, GPU call 40: tile-group stagger of the residual GEMMs on the 16x16x32 kernels (opt-in gemm_stagger), in the step
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c40"; mkdir -p "$OUT"
for r in 1 2 3; do for st in 0 1 45000 90000; do
  VITG_GEMM_STAGGER=$st timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/b.json" 2> "$OUT/b.err"
  python - "$OUT/b.json" $st <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"gemm_stagger={sys.argv[2]}: {d['value']} frames/s {d['ms_per_step']} ms  GEMM {d['roofline']['achieved']} TF/s parity {d['parity']['max_abs_dscore']}")
PY
done; done | tee "$OUT/step_ab.txt"
