#!/bin/bash
# round 4, GPU call 12: hd-128 attention with a balanced last round (full rounds of 256-query blocks + one launch of 128-query blocks): bit-identity
# tests, kernel A/B (profiling library: VITG_ATTN_WAVES=8 pins one launch), in-step A/B
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c12"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "balanced_last_round or attention" > "$OUT/pytest_attn.txt" 2>&1; tail -2 "$OUT/pytest_attn.txt"
for r in 1 2 3; do
  echo "## round $r balanced (default)"; VITG_PROFILING_LIB=1 timeout 300 python tools/attn_bench.py fp16 2>&1 | grep -E "llm"
  echo "## round $r one launch (attn_waves=8)"; VITG_PROFILING_LIB=1 timeout 300 python tools/attn_bench.py fp16 attn_waves=8 2>&1 | grep -E "llm"
done | tee "$OUT/attn128_balanced.txt"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o = d.get("roofline_other", {})
print(f"{sys.argv[2]:10s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  GEMM {d['roofline']['achieved']} TF/s  attn128 {o['attention_hd128']['avg_launch_ms']} ms {o['attention_hd128']['frac']}  parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
}
for r in 1 2 3; do one "bal_$r" VITG_PROFILING_LIB=1; one "one_$r" VITG_PROFILING_LIB=1 VITG_ATTN_WAVES=8; done | tee "$OUT/step_ab.txt"
one "product" X=1 | tee -a "$OUT/step_ab.txt"
