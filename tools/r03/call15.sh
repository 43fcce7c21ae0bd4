#!/bin/bash
# round 3, GPU call 15: hd-72 attention with the softmax row sums on the matrix pipe (constant pad chunk in V) against the VALU row sums
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c15"; mkdir -p "$OUT"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -k "attention" -x -q > "$OUT/pytest_attn.txt" 2>&1; tail -3 "$OUT/pytest_attn.txt"
for r in 1 2 3; do
  echo "## round $r: row sums on the matrix pipe (default)"; timeout 300 python tools/attn_bench.py fp16 2>&1 | grep -v amdgpu.ids | head -3
  echo "## round $r: VALU row sums (attn_ab=5)"; timeout 300 python tools/attn_bench.py fp16 attn_ab=5 2>&1 | grep -v amdgpu.ids | head -3
done | tee "$OUT/attn_ab.txt"
for r in 1 2; do
  for ab in 0 5; do
    VITG_ATTN_AB=$ab timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_ab${ab}_$r.json" 2> "$OUT/bench_ab${ab}_$r.err"
    python - "$OUT/bench_ab${ab}_$r.json" $ab <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"attn_ab={sys.argv[2]}: {d['value']} frames/s {d['ms_per_step']} ms  hd72 {d['roofline_other']['attention_hd72']} parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
  done
done | tee "$OUT/step_ab.txt"
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q > "$OUT/pytest_e2e.txt" 2>&1; tail -3 "$OUT/pytest_e2e.txt"
