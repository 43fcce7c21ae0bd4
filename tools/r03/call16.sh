#!/bin/bash
# round 3, GPU call 16: hd-72 attention taking its scores from the matrix pipe ready for exp2 (Q pre-scaled in the q projection's weights)
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c16"; mkdir -p "$OUT"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -k "attention" -x -q > "$OUT/pytest_attn.txt" 2>&1; tail -3 "$OUT/pytest_attn.txt"
for r in 1 2 3; do
  echo "## round $r: scores from the matrix pipe (variant=2, default kernel)"; timeout 300 python tools/attn_bench.py fp16 variant=2 2>&1 | grep -v amdgpu.ids | head -3
  echo "## round $r: pre-scaled Q, scale and max subtraction on the VALU (variant=2 attn_ab=6)"; timeout 300 python tools/attn_bench.py fp16 variant=2 attn_ab=6 2>&1 | grep -v amdgpu.ids | head -3
  echo "## round $r: plain (variant=0)"; timeout 300 python tools/attn_bench.py fp16 2>&1 | grep -v amdgpu.ids | head -3
done | tee "$OUT/attn_qs.txt"
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q > "$OUT/pytest_e2e.txt" 2>&1; tail -3 "$OUT/pytest_e2e.txt"
for r in 1 2; do
  for qp in 1 0; do
    VITG_Q_PRESCALE=$qp timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_qp${qp}_$r.json" 2> "$OUT/bench_qp${qp}_$r.err"
    python - "$OUT/bench_qp${qp}_$r.json" $qp <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d['roofline_other']['attention_hd72']
print(f"q_prescale={sys.argv[2]}: {d['value']} frames/s {d['ms_per_step']} ms  hd72 {h['achieved']} TF/s frac {h['frac']} avg {h['avg_launch_ms']} ms  parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
  done
done | tee "$OUT/step_ab.txt"
timeout 1200 python -m pytest tests/test_fullsize_parity.py -x -q -k "fp16" > "$OUT/pytest_fullsize.txt" 2>&1; tail -3 "$OUT/pytest_fullsize.txt"
