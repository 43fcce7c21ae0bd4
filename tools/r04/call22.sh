#!/bin/bash
# round 4, GPU call 22: package power / shader clock of the hd-128 attention kernel against the number of busy CUs (behind r04_experiments.md section 11)
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c22"; mkdir -p "$OUT"
timeout 600 python tools/power_attn_cu_count.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/power_attn_cu_count.txt"
