#!/bin/bash
# Build libvideoitg_hip.so for gfx950 (cross-compiles without a GPU).
#   build.sh [extra hipcc flags]               product library: shipped kernels only
#   build.sh --profiling [extra hipcc flags]   libvideoitg_hip_prof.so: + research GEMM variants and ablation switches
#                                              (include/videoitg_hip_profiling.h); loaded by tools/ only, never by the product
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRCS="device gemm gemm256 gemm_mi16 gemm_fp8 attention attention_sp attention_ab elementwise preprocess probe split"
OUT="$HERE/../libvideoitg_hip.so"
OBJ="$HERE/build"
DEFS=""
if [ "${1:-}" = "--profiling" ]; then
  shift
  SRCS="$SRCS gemm_w4 gemm_wf gemm_pe attention_w4"
  OUT="$HERE/../libvideoitg_hip_prof.so"
  OBJ="$HERE/build_prof"
  DEFS="-DVITG_PROFILING_BUILD=1"
fi
# build.sh --exp NAME [defs]: an EXPERIMENT flavour of the product library (libvideoitg_hip_exp_NAME.so, the A/B calls of tools/archive/r04_calls.sh.txt load it through
# VITG_LIB_PATH): the product sources with -DVITG_EXP=1 and the given -D switches; never built by __graft_entry__.build(), never loaded by default
if [ "${1:-}" = "--exp" ]; then
  NAME="$2"; shift 2
  OUT="$HERE/../libvideoitg_hip_exp_${NAME}.so"
  OBJ="$HERE/build_exp_${NAME}"
  DEFS="-DVITG_EXP=1"
fi
# VITG_CLEAN=1: compile every source from scratch (the default reuses objects newer than their sources and shared headers)
if [ "${VITG_CLEAN:-0}" != "0" ]; then rm -rf "$OBJ" "$OUT"; fi
mkdir -p "$OBJ"
pids=()
objs=()
compiled=""; reused=""
for f in $SRCS; do
  src="$HERE/$f.hip"; obj="$OBJ/$f.o"
  [ -f "$src" ] || continue
  objs+=("$obj")
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/common.h" -nt "$obj" ] || [ "$HERE/gemm_common.h" -nt "$obj" ] || [ "$HERE/attn_common.h" -nt "$obj" ] || [ "$HERE/../../include/videoitg_hip.h" -nt "$obj" ]; then
    extra=""
    [ "$f" = "attention_w4" ] && extra="-fno-slp-vectorize"      # its VALU stream is hand-placed in single-issue forms (see st_scale in the file)
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC $DEFS $extra -c "$src" -o "$obj" "$@" &
    pids+=($!)
    compiled="$compiled $f"
  else
    reused="$reused $f"
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
echo "compiled:${compiled:- (none)}  reused:${reused:- (none)}"
echo "built $OUT"
