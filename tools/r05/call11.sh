#!/bin/bash
# round 5, GPU call 11: 16-bit-output split-K for the Q / K|V projections of one context-parallel rank: kernel test, rank time with / without (same call), full-size parity at 8 / 4 / 2 ranks
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c11"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "splitk" > "$OUT/pytest_splitk.txt" 2>&1; tail -2 "$OUT/pytest_splitk.txt"
for rep in 1 2; do
  for v in 1 0; do
    echo "VITG_SPLITK16=$v (critical rank)"; VITG_SPLITK16=$v timeout 600 python tools/cp_rank_time.py 8 2>&1 | grep "rank-local"
    echo "VITG_SPLITK16=$v (rank 0)"; VITG_SPLITK16=$v VITG_CP_RANK=0 timeout 600 python tools/cp_rank_time.py 8 2>&1 | grep "rank-local"
  done
done | tee "$OUT/cp_rank8_ab.txt"
timeout 2400 python -m pytest tests/test_gpu_cp_fullsize.py tests/test_gpu_e2e.py -x -q -m gpu -s -k "context_parallel or multi_rank or bench_spawns" > "$OUT/pytest_cp.txt" 2>&1; grep -E "context-parallel x|passed|failed" "$OUT/pytest_cp.txt" | tail -12 | cut -c1-400
