#!/bin/bash
# round 5, GPU call 3: ablations of the one-wave-per-SIMD probe (what the K-tile's 3300 cycles are made of)
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c3"; mkdir -p "$OUT"
for v in 0 1 2 3 4 8 15 16 31 0; do
  for shape in "8192 8192 8192" "12800 37888 3584" "524288 4352 1152"; do
    timeout 300 tools/ubench/gemm_w4r_v$v $shape 10 0 2>&1 | grep TFLOP
  done
done | tee "$OUT/ablate.txt"
