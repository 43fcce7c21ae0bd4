#!/bin/bash
# round 4, GPU call 4: one rank of the 8-GPU step (stream launches vs graph replay), rocprofv3 kernel trace + PMC traffic + matrix-pipe
# counters of the bench workload on the current tree
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c4"; mkdir -p "$OUT"
for w in 8 4 2; do timeout 600 python tools/cp_rank_time.py $w 2>&1 | grep -E "world=" ; done | tee "$OUT/cp_rank_time.txt"
timeout 1500 bash tools/profile_step.sh r04 fp16 > "$OUT/profile_step.log" 2>&1; tail -5 "$OUT/profile_step.log"
timeout 900 bash tools/profile_mfma_util.sh r04 fp16 > "$OUT/profile_mfma.log" 2>&1; tail -12 "$OUT/profile_mfma.log"
cp gpurun_out/r04_* "$OUT/" 2>/dev/null || true
