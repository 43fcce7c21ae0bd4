#!/bin/bash
# round 4, GPU call 7: in-step A/B of non-temporal loads + stores in the tower's LayerNorm (experiment build)
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c7"; mkdir -p "$OUT"
EXP="$PWD/videoitg_amd/libvideoitg_hip_exp_x2.so"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o = d.get("roofline_other", {})
print(f"{sys.argv[2]:10s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  GEMM {d['roofline']['achieved']} TF/s  LN {o['layernorm']['avg_launch_ms']} ms {o['layernorm']['achieved']} GB/s  parity {d['parity']['max_abs_dscore']}")
PY
}
for r in 1 2 3; do
  one "exp0_$r" VITG_LIB_PATH="$EXP" VITG_EXP_LN_NT=0
  one "exp3_$r" VITG_LIB_PATH="$EXP" VITG_EXP_LN_NT=3
  one "exp1_$r" VITG_LIB_PATH="$EXP" VITG_EXP_LN_NT=1
done | tee "$OUT/step_ab.txt"
