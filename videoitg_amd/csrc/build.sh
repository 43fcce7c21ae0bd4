#!/bin/bash
# Build libvideoitg_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libvideoitg_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OBJ="$HERE/build"
mkdir -p "$OBJ"
pids=()
for f in device gemm gemm256 gemm_w4 gemm_wf gemm_pe attention attention_sp attention_ab elementwise preprocess probe; do
  src="$HERE/$f.hip"; obj="$OBJ/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/common.h" -nt "$obj" ] || [ "$HERE/gemm_common.h" -nt "$obj" ] || [ "$HERE/attn_common.h" -nt "$obj" ] || [ "$HERE/../../include/videoitg_hip.h" -nt "$obj" ]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o "$obj" "$@" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/device.o "$OBJ"/gemm.o "$OBJ"/gemm256.o "$OBJ"/gemm_w4.o "$OBJ"/gemm_wf.o "$OBJ"/gemm_pe.o "$OBJ"/attention.o "$OBJ"/attention_sp.o "$OBJ"/attention_ab.o "$OBJ"/elementwise.o "$OBJ"/preprocess.o "$OBJ"/probe.o
echo "built $OUT"
