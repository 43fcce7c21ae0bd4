#!/bin/bash
# Run ON THE GPU BOX: matrix-pipe utilisation (SQ counters, own pass, no trace domains) of the kernels of ONE operand-split forward of the headline workload.
#   gpurun --timeout 900 -- 'bash tools/r06/split_pmc.sh r06_c24'
set -uo pipefail
TAG="${1:-r06_split}"
REPO="$PWD"; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PCMD="python $REPO/tools/r06/split_step.py 1"
rm -rf /tmp/prof_s1 /tmp/prof_s2
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-include-regex "vitg" -d /tmp/prof_s1 -o split -- $PCMD > "$OUT/pmc1.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY --kernel-include-regex "vitg" -d /tmp/prof_s2 -o split -- $PCMD > "$OUT/pmc2.log" 2>&1
cd "$REPO"
python tools/pmc_kernels.py $(find /tmp/prof_s1 /tmp/prof_s2 -name "*.db") > "$OUT/split_pmc_mfma.txt" 2> "$OUT/pmc_kernels.err"
head -40 "$OUT/split_pmc_mfma.txt"
