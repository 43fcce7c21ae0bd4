#!/bin/bash
# round 5, GPU call 2: go / no-go of the one-wave-per-SIMD GEMM main loop (tools/ubench/gemm_w4r.hip) against the shipped kernel, same call
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c2"; mkdir -p "$OUT"
timeout 120 tools/ubench/gemm_w4r 1024 1024 1024 5 1 2>&1 | tee "$OUT/small.txt"
timeout 120 tools/ubench/gemm_w4r 512 768 192 5 1 2>&1 | tee -a "$OUT/small.txt"
timeout 900 python tools/r05/w4r_ab.py 2 2>&1 | grep -v amdgpu.ids | tee "$OUT/w4r_ab.txt"
