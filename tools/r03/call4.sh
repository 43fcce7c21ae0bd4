#!/bin/bash
# round 3, GPU call 4: new tests (res16 kernels, multi-image), MI16 per-shape A/B after the register fixes, tile preferences in-step, res16 gate
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c4"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -k "gemm or layernorm16 or pool16" -x -q > "$OUT/pytest_kernels.txt" 2>&1; tail -4 "$OUT/pytest_kernels.txt"
timeout 900 python -m pytest tests/test_gpu_e2e.py -k "golden or degenerate or mixed_batch or residual16" -x -q > "$OUT/pytest_e2e.txt" 2>&1; tail -4 "$OUT/pytest_e2e.txt"
timeout 900 python tools/gemm_mi_ab.py fp16 3 > "$OUT/gemm_mi_ab_fp16.txt" 2>&1
cat "$OUT/gemm_mi_ab_fp16.txt"
run() {  # mi pref tag [env]
  env VITG_GEMM_MI=$1 VITG_GEMM_PREF=$2 ${4:-X=1} timeout 600 python bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_$3.json" 2> "$OUT/bench_$3.err"
  python - <<PY
import json
r = json.loads(open("$OUT/bench_$3.json").read().strip().splitlines()[-1])
o = r.get("roofline_other", {})
print("$3 (mi $1 pref $2 ${4:-}):", r["value"], "frames/s", r["ms_per_step"], "ms  GEMM", r["roofline"]["achieved"], "TF/s frac", r["roofline"]["frac"], "parity", r["parity"]["max_abs_dscore"], r["parity"]["pass"],
      "LN", o.get("layernorm", {}).get("achieved"), "patchify", o.get("patchify", {}).get("achieved"), "rope", o.get("rope", {}).get("achieved"), "pool", o.get("token_pool", {}).get("achieved"), flush=True)
PY
}
run 32 0 mi32_a
run 16 0 mi16_auto_a
run 16 200 mi16_o9
run 16 100 mi16_o6
run 16 20 mi16_f9
run 16 30 mi16_f10
run 16 10 mi16_f6
run 16 0 mi16_auto_b
run 32 0 mi32_b
run 16 0 mi16_res16 VITG_RES16=1
run 32 0 mi32_res16 VITG_RES16=1
VITG_GEMM_MI=16 timeout 1500 python tools/res16_gate.py > "$OUT/res16_gate_mi16.txt" 2>&1; cat "$OUT/res16_gate_mi16.txt"
