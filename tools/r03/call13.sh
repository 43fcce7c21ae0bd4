#!/bin/bash
# round 3, GPU call 13: stream-K for the per-rank o / down projections (tests, small-M table, rank step with / without), training-mode pooling test
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c13"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "streamk or splitk" -x -q > "$OUT/pytest_sk.txt" 2>&1; tail -3 "$OUT/pytest_sk.txt"
timeout 900 python -m pytest tests/test_gpu_e2e.py -k "training_mode or context_parallel" -x -q > "$OUT/pytest_e2e.txt" 2>&1; tail -3 "$OUT/pytest_e2e.txt"
timeout 900 python tools/gemm_small_m.py fp16 2>&1 | grep -v amdgpu.ids | grep -E " o | down " | tee "$OUT/gemm_small_m.txt"
for r in 1 2; do for skv in 1 0; do echo "VITG_STREAMK=$skv"; VITG_STREAMK=$skv timeout 600 python tools/cp_rank_time.py 8 2>&1 | tail -2 | head -1; done; done | tee "$OUT/cp_rank_time.txt"
for w in 4 2; do timeout 600 python tools/cp_rank_time.py $w 2>&1 | tail -2 | head -1; done | tee -a "$OUT/cp_rank_time.txt"
