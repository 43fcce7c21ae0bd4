#!/bin/bash
# round 4, GPU call 29: fp8 tower GEMMs on the tower's 16-bit residual stream (vitg_gemm_fp8_res16 / vitg_norm_quant_fp8_16): kernel tests, the fp8 test
# file, A/B of the all-eight and tower-only fp8 jobs against the fp32 stream (VITG_FP8_R16=0)
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c29"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu -k "res16 or 16_bit_rows" 2>&1 | tail -3 | tee "$OUT/pytest_new.txt"
timeout 1800 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu > "$OUT/pytest_fp8.txt" 2>&1; tail -3 "$OUT/pytest_fp8.txt"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --fp8 --fp8-policy all --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:12s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  fp8 GEMM {r['achieved']} TF/s = {r['frac']} ({r['launches']} launches, {r['avg_launch_ms']} ms)  max dscore {d['parity']['max_abs_dscore']}  top32 kept {d['parity']['topk_ladder']['top32']['kept']}")
PY
}
for r in 1 2 3; do
  one "fp32stream_$r" VITG_FP8_R16=0
  one "stream16_$r" VITG_FP8_R16=1
done | tee "$OUT/step_ab.txt"
