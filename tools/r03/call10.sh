#!/bin/bash
# round 3, GPU call 10: 192-row wave tiles in intervals of 6 (MC = 6) against 4; context-parallel rank with / without the K|V-first split
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c10"; mkdir -p "$OUT"
bash tools/r03/ab_mc6.sh 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_mc6.txt"
for r in 1 2; do for ov in 1 0; do echo "VITG_CP_OVERLAP=$ov"; VITG_CP_OVERLAP=$ov timeout 600 python tools/cp_rank_time.py 8 2>&1 | tail -2 | head -1; done; done | tee "$OUT/cp_rank_time.txt"
