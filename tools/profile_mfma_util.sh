#!/bin/bash
# Run ON THE GPU BOX: matrix-pipe utilisation of the MFMA kernels of one bench step from SQ counters (own pass, library kernels only).
#   gpurun --timeout 900 -- 'bash tools/profile_mfma_util.sh r02 fp16'
set -uo pipefail
TAG="${1:-r02}"; DT="${2:-fp16}"
REPO="$PWD"; OUT="$PWD/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PCMD="python $REPO/bench.py --steps 1 --warmup 0 --dtype $DT --no-cpu-baseline --no-secondary"
rm -rf /tmp/prof_m1 /tmp/prof_m2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-include-regex "vitg" -d /tmp/prof_m1 -o bench -- $PCMD > /dev/null 2> "$OUT/${TAG}_prof_m1.err"
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY --kernel-include-regex "vitg" -d /tmp/prof_m2 -o bench -- $PCMD > /dev/null 2> "$OUT/${TAG}_prof_m2.err"
cd "$REPO"
python tools/pmc_kernels.py $(find /tmp/prof_m1 /tmp/prof_m2 -name "*.db") > "$OUT/${TAG}_pmc_mfma_${DT}.txt" 2>> "$OUT/${TAG}_prof_m1.err"
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_BUSY|SQ_WAVE_CYCLES" | head -30 > "$OUT/${TAG}_counters_avail.txt"
head -40 "$OUT/${TAG}_pmc_mfma_${DT}.txt"
