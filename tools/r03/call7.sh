#!/bin/bash
# round 3, GPU call 7: the whole GPU suite on the final kernels; rocprofv3 kernel trace + HBM-traffic PMC passes of the bench step (profiles/r03_*);
# one rank of the 8-way context-parallel step under rocprofv3 (where the per-rank LLM time goes at M = 1660)
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c7"; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1; tail -8 "$OUT/pytest_gpu.txt"
timeout 1500 bash tools/profile_step.sh r03 fp16 > "$OUT/profile_step.log" 2>&1; tail -5 "$OUT/profile_step.log"
cp gpurun_out/r03_kernel_stats_fp16.md gpurun_out/r03_pmc_traffic_fp16.md gpurun_out/r03_gemm_traffic_fp16.json gpurun_out/r03_prof_bench_fp16.json "$OUT/" 2>/dev/null
export TMPDIR=/tmp; REPO="$PWD"; cd /tmp; rm -rf /tmp/prof_cp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_cp -o cp -- python $REPO/tools/cp_rank_time.py 8 > "$OUT/cp_rank_time_prof.log" 2>&1
DB=$(find /tmp/prof_cp -name "*.db" | head -1); python $REPO/tools/rocpd_summary.py "$DB" > "$OUT/cp_rank8_kernel_stats.md" 2>> "$OUT/cp_rank_time_prof.log"
cd "$REPO"; tail -3 "$OUT/cp_rank_time_prof.log"; head -40 "$OUT/cp_rank8_kernel_stats.md"
for w in 8; do timeout 600 python tools/cp_rank_time.py $w 2>&1 | tail -2; done | tee "$OUT/cp_rank_time.txt"
