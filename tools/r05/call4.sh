#!/bin/bash
# round 5, GPU call 4: the probe with per-wave staggered deposit / request slots (SCHED 1) against the lockstep schedule (SCHED 0) and its ablations
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c4"; mkdir -p "$OUT"
timeout 120 tools/ubench/gemm_w4r_s1v0 1024 1024 1024 5 1 2>&1 | tee "$OUT/ab.txt"
timeout 120 tools/ubench/gemm_w4r_s1v0 512 768 192 5 1 2>&1 | tee -a "$OUT/ab.txt"
for b in s0v0 s1v0 s1v1 s1v2 s1v16 s0v0 s1v0; do
  for shape in "8192 8192 8192" "12800 37888 3584" "524288 4352 1152"; do
    timeout 300 tools/ubench/gemm_w4r_$b $shape 10 0 2>&1 | grep TFLOP
  done
done | tee -a "$OUT/ab.txt"
