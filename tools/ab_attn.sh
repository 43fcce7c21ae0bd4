#!/bin/bash
# A/B attention variants: for each abl/lib_*.so, swap it in as the product library and run tools/attn_bench.py
cd "$(dirname "$0")/.."
cp videoitg_amd/libvideoitg_hip.so /tmp/lib_orig.so
for l in abl/lib_*.so; do
  cp "$l" videoitg_amd/libvideoitg_hip.so
  echo "== $l"; python tools/attn_bench.py "$@" 2>&1 | tail -3
done
cp /tmp/lib_orig.so videoitg_amd/libvideoitg_hip.so
