#!/bin/bash
# round 4, GPU call 27: 16-bit-residual GEMMs that touch the NEXT tile's residual lines during their main loop (experiment build respf): does the
# accumulator preload burst get shorter when its lines come from the L2 / Infinity Cache?  stand-alone and in the step
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c27"; mkdir -p "$OUT"
EXP="$PWD/videoitg_amd/libvideoitg_hip_exp_respf.so"
for r in 1 2; do
  python tools/experiments/res16_seq_tn_ab.py
  VITG_LIB_PATH="$EXP" python tools/experiments/res16_seq_tn_ab.py
done 2>&1 | grep -v amdgpu.ids | tee "$OUT/res16_seq_tn_ab.txt"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:12s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  GEMM {d['roofline']['achieved']} TF/s {d['roofline']['avg_launch_ms']} ms  parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
}
for r in 1 2; do
  one "grouped_$r" X=1
  one "respf_$r" VITG_LIB_PATH="$EXP"
done | tee "$OUT/step_ab.txt"
