#!/bin/bash
# round 4, GPU call 8: "serpentine" launch order (successive big kernels of the tower walk their tile lists in alternating directions, so that a
# kernel starts on the rows its predecessor wrote last - still in the Infinity Cache); experiment build, in-step A/B
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c8"; mkdir -p "$OUT"
EXP="$PWD/videoitg_amd/libvideoitg_hip_exp_x2.so"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o = d.get("roofline_other", {})
print(f"{sys.argv[2]:10s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  GEMM {d['roofline']['achieved']} TF/s  LN {o['layernorm']['avg_launch_ms']} ms  attn72 {o['attention_hd72']['avg_launch_ms']} ms  parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
}
for r in 1 2 3; do
  one "serp0_$r" VITG_LIB_PATH="$EXP" VITG_EXP_SERP=0
  one "serp1_$r" VITG_LIB_PATH="$EXP" VITG_EXP_SERP=1
  one "serp3_$r" VITG_LIB_PATH="$EXP" VITG_EXP_SERP=3
done | tee "$OUT/step_ab.txt"
