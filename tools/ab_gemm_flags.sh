#!/bin/bash
# (round 4: VITG_GEMM_FLAGS / VITG_GEMM_GROUP / VITG_GEMM_STAGGER are research knobs of the profiling library - the product library ignores them;
#  this script is the record of the round-1..3 A/Bs and needs VITG_PROFILING_LIB=1 to repeat them)
# in-step A/B of the tile-shape cost-model switches (VITG_GEMM_FLAGS) and tile-order group (VITG_GEMM_GROUP), fp16 headline workload
for rep in 1 2; do
for f in 0 4096 2048 256 8192 16384 16777216; do
  VITG_GEMM_FLAGS=$f python bench.py --steps 4 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags', $f, 'ms', r['ms_per_step'], 'gemm TF/s', r['roofline']['achieved'], 'parity', r['parity']['max_abs_dscore'])"
done
for g in 2 8; do
  VITG_GEMM_GROUP=$g python bench.py --steps 4 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group', $g, 'ms', r['ms_per_step'], 'gemm TF/s', r['roofline']['achieved'])"
done
done
