#!/bin/bash
# round 3, GPU call 11: hd-128 attention with PV on 16x16x32 (tests, A/B against the round-2 form), SwiGLU intervals of 6, step
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c11"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -k "attention" -x -q > "$OUT/pytest_attn.txt" 2>&1; tail -3 "$OUT/pytest_attn.txt"
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_graphs.py -x -q > "$OUT/pytest_e2e.txt" 2>&1; tail -3 "$OUT/pytest_e2e.txt"
for r in 1 2 3; do
  echo "## round $r: hd128 PV 16x16x32 (default)"; timeout 300 python tools/attn_bench.py fp16 2>&1 | grep "llm"
  echo "## round $r: hd128 PV 32x32x16 (attn_sp=3)"; timeout 300 python tools/attn_bench.py fp16 attn_sp=3 2>&1 | grep "llm"
done | tee "$OUT/attn128_ab.txt"
run() {  # tag env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - <<PY
import json
r = json.loads(open("$OUT/bench_$tag.json").read().strip().splitlines()[-1])
o = r.get("roofline_other", {})
print("$tag ($*):", r["value"], "frames/s", r["ms_per_step"], "ms  GEMM", r["roofline"]["achieved"], "TF/s frac", r["roofline"]["frac"], "parity", r["parity"]["max_abs_dscore"], r["parity"]["pass"],
      "a72", o.get("attention_hd72", {}).get("achieved"), "a128", o.get("attention_hd128", {}).get("achieved"), o.get("attention_hd128", {}).get("frac"), flush=True)
PY
}
run pv16_a X=1
run sp_pv32_a VITG_ATTN_SP=3
run pv16_b X=1
run sp_pv32_b VITG_ATTN_SP=3
