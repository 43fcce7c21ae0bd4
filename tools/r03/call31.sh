#!/bin/bash
# round 3, GPU call 31: the profiling-library test file (skipped in the product suite) after the shared-epilogue changes
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c31"; mkdir -p "$OUT"
VITG_PROFILING_LIB=1 timeout 1800 python -m pytest tests/test_gpu_profiling_lib.py -x -q > "$OUT/pytest_prof.txt" 2>&1; tail -3 "$OUT/pytest_prof.txt"
