#!/bin/bash
# round 4, GPU call 5: the hardware questions behind VERDICT r3 item 3 (half-K MFMA cost, polynomial exp2), the extended full-size
# context-parallel test (b0 at 8 ranks, c1 at 4 ranks)
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c5"; mkdir -p "$OUT"
timeout 300 tools/ubench/softmax_levers 2>&1 | tee "$OUT/softmax_levers.txt"
timeout 1500 python -m pytest tests/test_gpu_cp_fullsize.py -x -q -m gpu -s > "$OUT/pytest_cp.txt" 2>&1
grep -E "context-parallel x|passed|failed|Error" "$OUT/pytest_cp.txt" | cut -c1-400 | tail -12
