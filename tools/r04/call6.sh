#!/bin/bash
# round 4, GPU call 6: cache-policy A/B of the tower's LayerNorm (experiment build), new-tree sanity
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c6"; mkdir -p "$OUT"
EXP="$PWD/videoitg_amd/libvideoitg_hip_exp_x2.so"
for r in 1 2 3; do for nt in 0 1 2 3; do echo -n "nt=$nt "; VITG_LIB_PATH="$EXP" VITG_EXP_LN_NT=$nt timeout 120 python tools/ln16_bench.py 2>&1 | tail -1; done; done | tee "$OUT/ln16_nt.txt"
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm or patch or norm" 2>&1 | tail -2
