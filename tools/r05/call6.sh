#!/bin/bash
# round 5, GPU call 6: per-shape A/B of the 320-row tile on one rank's decoder GEMMs; heavy-tailed fixtures on both tower streams; fp8 error anatomy
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c6"; mkdir -p "$OUT"
timeout 600 python tools/r05/cp_gemm_ab.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/cp_gemm_ab.txt"
timeout 1500 python -m pytest tests/test_fullsize_parity.py -q -m gpu -s -k "heavy_tailed" > "$OUT/pytest_heavy.txt" 2>&1; grep -E "heavy-tailed|passed|failed|Error" "$OUT/pytest_heavy.txt" | tail -12
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "engine_default" 2>&1 | grep -E "fp8 tower|passed|failed" | tee "$OUT/pytest_engine_default.txt"
timeout 1500 python tools/fp8_error_floor.py gpu c2_512x448 c2p_512x448 c2o_512x448 2>&1 | grep -v amdgpu.ids | tee "$OUT/fp8_error_floor.txt"
