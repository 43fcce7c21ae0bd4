#!/bin/bash
# (needs the by-work attention of commit b76ba79: git apply tools/experiments/attention_by_work.patch, rebuild)
# round 4, GPU call 16: hd-128 attention cut by WORK (vitg_attention_ws, attention_sp.hip LEAN): kernel tests, stand-alone A/B on the one-GPU and the
# per-rank shapes, A/B inside one rank of the 8-way step and inside the one-GPU step
set -uo pipefail
OUT="$PWD/gpurun_out/${CALL_TAG:-r04_c16}"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > "$OUT/pytest_attention.txt" 2>&1; tail -3 "$OUT/pytest_attention.txt"
timeout 600 python tools/experiments/attn_lean_ab.py 3 2>&1 | tee "$OUT/attn_lean_ab.txt"
for r in 1 2; do
  for v in 0 1; do
    echo "VITG_LEAN_ATTN=$v round $r"
    VITG_LEAN_ATTN=$v timeout 600 python tools/cp_rank_time.py 8 2>&1 | grep "world="
  done
done | tee "$OUT/cp_rank_time_ab.txt"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
a = d["roofline_other"]["attention_hd128"]
print(f"{sys.argv[2]:10s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  GEMM {d['roofline']['achieved']} TF/s  hd128 {a['avg_launch_ms']} ms {a['frac']}  parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
}
for r in 1 2; do
  one "classic_$r" VITG_LEAN_ATTN=0
  one "bywork_$r" VITG_LEAN_ATTN=1
done | tee "$OUT/step_ab.txt"
