#!/bin/bash
# round 3, GPU call 2: gemm_mi16.hip - correctness (kernel tests with every tile shape on both MFMA shapes), per-shape A/B, step A/B
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c2"; mkdir -p "$OUT"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -k "gemm" -x -q > "$OUT/pytest_gemm.txt" 2>&1
tail -5 "$OUT/pytest_gemm.txt"
timeout 900 python tools/gemm_mi_ab.py fp16 3 > "$OUT/gemm_mi_ab_fp16.txt" 2>&1
cat "$OUT/gemm_mi_ab_fp16.txt"
for r in 1 2; do
  for mi in 32 16; do
    VITG_GEMM_MI=$mi timeout 600 python bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_mi${mi}_r$r.json" 2> "$OUT/bench_mi${mi}_r$r.err"
    python - <<PY
import json
r = json.loads(open("$OUT/bench_mi${mi}_r$r.json").read().strip().splitlines()[-1])
print("mi$mi run $r:", r["value"], "frames/s", r["ms_per_step"], "ms  GEMM", r["roofline"]["achieved"], "TF/s frac", r["roofline"]["frac"], "parity", r["parity"]["max_abs_dscore"], r["parity"]["pass"])
PY
  done
done
timeout 600 python tools/gemm_mi_ab.py bf16 2 > "$OUT/gemm_mi_ab_bf16.txt" 2>&1
cat "$OUT/gemm_mi_ab_bf16.txt"
