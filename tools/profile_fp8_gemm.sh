#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the fp8 GEMM (and the 16-bit GEMM beside it) on the shapes of the step.
set -uo pipefail
REPO="$PWD"; OUT="$PWD/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_f8a /tmp/prof_f8b
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-include-regex "gemm" -d /tmp/prof_f8a -o b -- python $REPO/tools/gemm_fp8_bench.py > /dev/null 2> "$OUT/r02_f8a.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-include-regex "gemm" -d /tmp/prof_f8b -o b -- python $REPO/tools/gemm_fp8_bench.py > /dev/null 2> "$OUT/r02_f8b.err"
cd "$REPO"
python tools/pmc_kernels.py $(find /tmp/prof_f8a /tmp/prof_f8b -name "*.db") > "$OUT/r02_pmc_fp8_gemm.txt" 2>> "$OUT/r02_f8a.err"
