#!/bin/bash
# round 4, GPU call 18: LayerNorm over 16-bit rows with FOUR rows per wave / 16 lanes per row (experiment build ln16r4) against the shipped one row per wave:
# stand-alone rate beside this box's copy rate, launch geometry, the kernel tests, A/B in the step
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c18"; mkdir -p "$OUT"
EXP="$PWD/videoitg_amd/libvideoitg_hip_exp_ln16r4.so"
{
python tools/ln16_bench.py --check
VITG_LIB_PATH="$EXP" python tools/ln16_bench.py --check
for cap in 2048 4096 8192 16384 32768; do for blk in 256 512; do VITG_LN16R4_CAP=$cap VITG_LN16R4_BLOCK=$blk VITG_LIB_PATH="$EXP" python tools/ln16_bench.py; done; done
python tools/ln16_bench.py
} 2>&1 | grep -v amdgpu.ids | tee "$OUT/ln16_bench.txt"
VITG_LIB_PATH="$EXP" timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm16 or ln16" 2>&1 | tail -2 | tee "$OUT/pytest_ln16.txt"
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
a = d["roofline_other"]["layernorm"]
print(f"{sys.argv[2]:10s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  GEMM {d['roofline']['achieved']} TF/s  LN {a['avg_launch_ms']} ms {a['achieved']} GB/s  parity {d['parity']['max_abs_dscore']} {d['parity']['pass']}")
PY
}
for r in 1 2 3; do
  one "one_row_$r" X=1
  one "four_rows_$r" VITG_LIB_PATH="$EXP"
done | tee "$OUT/step_ab.txt"
