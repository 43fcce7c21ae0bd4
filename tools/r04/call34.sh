#!/bin/bash
# round 4, GPU call 34: wide tiles for the short-K fp8 GEMMs of the tower (256 x 384 plain, 384 x 256 MX-producing): the fp8 test file, per-shape rates,
# A/B of the fp8 job (policy seven) against VITG_FP8_WIDE=0
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c34"; mkdir -p "$OUT"
timeout 1800 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu > "$OUT/pytest_fp8.txt" 2>&1; tail -3 "$OUT/pytest_fp8.txt"
for v in 0 1; do echo "VITG_FP8_WIDE=$v"; VITG_FP8_WIDE=$v timeout 600 python tools/gemm_fp8_bench.py 2>&1 | grep -E "vit_qkv|vit_fc1"; done | tee "$OUT/gemm_fp8_bench.txt"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --fp8 --fp8-policy seven --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:10s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  fp8 GEMM {r['achieved']} TF/s = {r['frac']} ({r['launches']} launches)  max dscore {d['parity']['max_abs_dscore']}  top32 kept {d['parity']['topk_ladder']['top32']['kept']}")
PY
}
for r in 1 2 3; do
  one "tile256_$r" VITG_FP8_WIDE=0
  one "wide_$r" VITG_FP8_WIDE=1
done | tee "$OUT/fp8_job_ab.txt"
