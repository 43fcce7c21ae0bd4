#!/bin/bash
# round 3, GPU call 33: grid / block sweep of the 16-bit LayerNorm
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c33"; mkdir -p "$OUT"
for r in 1 2; do
for blk in 256 512 128 1024; do for cap in 4096 8192 16384 32768 65536 1000000; do
  VITG_LN16_CAP=$cap VITG_LN16_BLOCK=$blk timeout 120 python tools/ln16_bench.py 2>&1 | tail -1
done; done; done | tee "$OUT/ln16_sweep.txt"
