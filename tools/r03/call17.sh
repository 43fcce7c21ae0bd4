#!/bin/bash
# round 3, GPU call 17: planted-signal fixtures (16-bit path, fp8 tiers, the per-GEMM policy sweep on the planted case)
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c17"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_fullsize_parity.py -x -q -s -k "p_" > "$OUT/pytest_fullsize_planted.txt" 2>&1; grep -E "^full|passed|failed" "$OUT/pytest_fullsize_planted.txt"
timeout 1500 python -m pytest tests/test_gpu_fp8.py -x -q -s -k "planted" > "$OUT/pytest_fp8_planted.txt" 2>&1; grep -E "^fp8|passed|failed" "$OUT/pytest_fp8_planted.txt"
timeout 1800 python tools/fp8_policy_sweep.py c2p_512x448 2>&1 | grep -v amdgpu.ids | tee "$OUT/fp8_policy_sweep_planted.txt"
