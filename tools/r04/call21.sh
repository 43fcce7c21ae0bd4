#!/bin/bash
# round 4, GPU call 21: tower qkv projection written head-major (vitg_gemm_heads16): kernel test, full-size parity + e2e files, A/B in the step
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c21"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "heads16" 2>&1 | tail -3 | tee "$OUT/pytest_heads16.txt"
timeout 1500 python -m pytest tests/test_fullsize_parity.py tests/test_gpu_e2e.py tests/test_gpu_graphs.py -x -q -m gpu > "$OUT/pytest_e2e.txt" 2>&1; tail -3 "$OUT/pytest_e2e.txt"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
a = d["roofline_other"]["attention_hd72"]
print(f"{sys.argv[2]:12s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  GEMM {d['roofline']['achieved']} TF/s {d['roofline']['avg_launch_ms']} ms  hd72 {a['avg_launch_ms']} ms {a['frac']}  parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
}
for r in 1 2 3; do
  one "rowmajor_$r" VITG_HEAD_MAJOR=0
  one "headmajor_$r" VITG_HEAD_MAJOR=1
done | tee "$OUT/step_ab.txt"
