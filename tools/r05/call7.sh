#!/bin/bash
# round 5, GPU call 7: where the heavy-tailed fixtures' error enters (stage by stage against the oracle on the box's host cores)
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c7"; mkdir -p "$OUT"
timeout 1500 python tools/heavy_tail_stage_diff.py 16 224 2>&1 | grep -v amdgpu.ids | tee "$OUT/stage_diff.txt"
