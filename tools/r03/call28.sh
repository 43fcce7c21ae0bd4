#!/bin/bash
# round 3, GPU call 28: rocprofv3 kernel trace + PMC traffic of the final tree, per-rank step at 8 / 4 / 2 ranks, vendor comparison, small-M table
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c28"; mkdir -p "$OUT"
bash tools/profile_step.sh r03 fp16 > "$OUT/profile_step.log" 2>&1; tail -5 "$OUT/profile_step.log"
cp gpurun_out/r03_kernel_stats_fp16.md gpurun_out/r03_pmc_traffic_fp16.md gpurun_out/r03_gemm_traffic_fp16.json gpurun_out/r03_prof_bench_fp16.json "$OUT/" 2>/dev/null
for w in 8 8 4 2; do timeout 600 python tools/cp_rank_time.py $w 2>&1 | tail -2 | head -1; done | tee "$OUT/cp_rank_time.txt"
timeout 900 python tools/gemm_vs_vendor.py fp16 2>&1 | grep -v amdgpu.ids | tee "$OUT/gemm_vs_vendor_fp16.txt"
timeout 900 python tools/gemm_mi_ab.py fp16 1 2>&1 | grep -v amdgpu.ids | tee "$OUT/gemm_shapes.txt"
