#!/bin/bash
# round 3, GPU call 3: gemm_mi16.hip after the epilogue spill fix - per-shape A/B, then tile-shape preferences decided INSIDE the step
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c3"; mkdir -p "$OUT"
timeout 900 python tools/gemm_mi_ab.py fp16 3 > "$OUT/gemm_mi_ab_fp16.txt" 2>&1
cat "$OUT/gemm_mi_ab_fp16.txt"
run() {  # mi pref tag
  VITG_GEMM_MI=$1 VITG_GEMM_PREF=$2 timeout 600 python bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_$3.json" 2> "$OUT/bench_$3.err"
  python - <<PY
import json
r = json.loads(open("$OUT/bench_$3.json").read().strip().splitlines()[-1])
print("$3 (mi $1 pref $2):", r["value"], "frames/s", r["ms_per_step"], "ms  GEMM", r["roofline"]["achieved"], "TF/s frac", r["roofline"]["frac"], "parity", r["parity"]["max_abs_dscore"], r["parity"]["pass"], flush=True)
PY
}
run 32 0 mi32_a
run 16 0 mi16_auto_a
run 16 200 mi16_o9
run 16 300 mi16_o10
run 16 100 mi16_o6
run 16 20 mi16_f9
run 16 30 mi16_f10
run 16 10 mi16_f6
run 16 1 mi16_s6
run 16 0 mi16_auto_b
run 32 0 mi32_b
