#!/bin/bash
# round 5, GPU call 12: kernel trace of ONE context-parallel rank's step (critical rank, 1660 rows) - where its 32 ms of decoder go
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c12"; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "splitk" 2>&1 | tail -1
export TMPDIR=/tmp; REPO="$PWD"; cd /tmp; rm -rf /tmp/prof_cp
rocprofv3 --kernel-trace --stats -d /tmp/prof_cp -o cp -- python $REPO/tools/cp_rank_time.py 8 > "$OUT/cp_under_rocprof.txt" 2>&1
DB=$(find /tmp/prof_cp -name "*.db" | head -1)
cd "$REPO"; python tools/rocpd_summary.py "$DB" > "$OUT/cp_rank8_kernel_stats.md" 2>> "$OUT/cp_under_rocprof.txt"
grep vitg "$OUT/cp_rank8_kernel_stats.md" | head -30 | cut -c1-170
