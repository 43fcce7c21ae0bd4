#!/bin/bash
# round 3, GPU call 1: MFMA shape vs power cap; output-store cache policy and XCD round rendezvous on the tower GEMMs (time + FETCH_SIZE)
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c1"; mkdir -p "$OUT"
export TMPDIR=/tmp
REPO="$PWD"
timeout 300 tools/ubench/mfma_shape_power > "$OUT/mfma_shape_power.txt" 2>&1
timeout 900 python tools/gemm_l2_ab.py 5 > "$OUT/gemm_l2_ab.txt" 2>&1
cd /tmp
for sh in vit_fc1 vit_qkv; do
  for v in base sc1 sync sync+sc1; do
    rm -rf /tmp/pf
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gemm256" -d /tmp/pf -o x -- python $REPO/tools/gemm_l2_ab.py pmc $v $sh > /dev/null 2> "$OUT/pmc_${sh}_${v}.err"
    DB=$(find /tmp/pf -name "*.db" | head -1)
    echo "## $sh $v FETCH" >> "$OUT/pmc.txt"; python $REPO/tools/pmc_kernels.py $DB >> "$OUT/pmc.txt" 2>&1
  done
  for v in base sc1; do
    rm -rf /tmp/pf
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "gemm256" -d /tmp/pf -o x -- python $REPO/tools/gemm_l2_ab.py pmc $v $sh > /dev/null 2>> "$OUT/pmc_${sh}_${v}.err"
    DB=$(find /tmp/pf -name "*.db" | head -1)
    echo "## $sh $v WRITE" >> "$OUT/pmc.txt"; python $REPO/tools/pmc_kernels.py $DB >> "$OUT/pmc.txt" 2>&1
  done
done
cd "$REPO"
timeout 600 python bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_head.json" 2> "$OUT/bench_head.err"
tail -n 40 "$OUT/mfma_shape_power.txt" "$OUT/gemm_l2_ab.txt" "$OUT/pmc.txt"
