#!/bin/bash
# round 3, GPU call 5: full GPU suite on the new defaults (16x16x32 GEMMs, fp16 tower stream), per-shape A/B incl. res16 shapes, in-step A/B of
# the remaining choices, one rank of the 8-way context-parallel step, fp8 per-GEMM policies, the full default bench line
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c5"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1; tail -6 "$OUT/pytest_gpu.txt"
timeout 900 python tools/gemm_mi_ab.py fp16 3 > "$OUT/gemm_mi_ab_fp16.txt" 2>&1
cat "$OUT/gemm_mi_ab_fp16.txt"
run() {  # tag env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - <<PY
import json
r = json.loads(open("$OUT/bench_$tag.json").read().strip().splitlines()[-1])
o = r.get("roofline_other", {})
print("$tag ($*):", r["value"], "frames/s", r["ms_per_step"], "ms  GEMM", r["roofline"]["achieved"], "TF/s frac", r["roofline"]["frac"], "parity", r["parity"]["max_abs_dscore"], r["parity"]["pass"],
      "LN", o.get("layernorm", {}).get("achieved"), "a72", o.get("attention_hd72", {}).get("achieved"), "a128", o.get("attention_hd128", {}).get("achieved"), flush=True)
PY
}
run default_a X=1
run swiglu_mi32 VITG_GEMM_MI_MODES=3
run resid_mi32 VITG_GEMM_MI_MODES=5
run o6 VITG_GEMM_PREF=100
run o10 VITG_GEMM_PREF=300
run f9 VITG_GEMM_PREF=20
run f10 VITG_GEMM_PREF=30
run default_b X=1
run r2_kernels VITG_GEMM_MI=32 VITG_RES16=0
for w in 8 4 2; do timeout 600 python tools/cp_rank_time.py $w 2>&1 | tail -3; done | tee "$OUT/cp_rank_time.txt"
timeout 1500 python tools/fp8_policy_sweep.py > "$OUT/fp8_policy_sweep.txt" 2>&1; cat "$OUT/fp8_policy_sweep.txt"
timeout 1500 python bench.py > "$OUT/bench_full_default.json" 2> "$OUT/bench_full_default.err"; tail -c 3000 "$OUT/bench_full_default.json"
