#!/bin/bash
# (round 4: VITG_GEMM_FLAGS / VITG_GEMM_GROUP / VITG_GEMM_STAGGER are research knobs of the profiling library - the product library ignores them;
#  this script is the record of the round-1..3 A/Bs and needs VITG_PROFILING_LIB=1 to repeat them)
# same-box A/B of the tile-group de-phasing for the fp32-residual GEMMs (gemm_stagger): per shape, then in the step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for r in 1 2; do
  for s in 0 1 50000 120000; do
    echo "#### gemm_stagger=$s run $r"
    VITG_GEMM_STAGGER=$s python tools/gemm_fp8_bench.py 2>&1 | grep -E "vit_out|vit_fc2|llm_o|llm_down" | cut -c1-75
  done
done
for r in 1 2; do
  for s in 0 1; do
    echo "#### step gemm_stagger=$s run $r"
    VITG_GEMM_STAGGER=$s python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['achieved'], r['parity']['max_abs_dscore'])"
  done
done
