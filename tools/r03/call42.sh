#!/bin/bash
# round 3, GPU call 42: residual-form bias added in one pass behind the main loop (bit27 = per quad in the stages; bit27 also moves the 16-bit-output bias back into the epilogue)
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c42"; mkdir -p "$OUT"
for r in 1 2 3; do for e in 0 67108864; do
  VITG_GEMM_FLAGS=$e timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/b.json" 2> "$OUT/b.err"
  python - "$OUT/b.json" $e <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"gemm_flags={sys.argv[2]}: {d['value']} frames/s {d['ms_per_step']} ms  GEMM {d['roofline']['achieved']} TF/s parity {d['parity']['max_abs_dscore']}")
PY
done; done | tee "$OUT/step_ab.txt"
