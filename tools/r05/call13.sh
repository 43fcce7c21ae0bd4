#!/bin/bash
# round 5, GPU call 13: the 352x256 tile (cfg 14): every GEMM test under every forced tile, per-shape A/B at M = 1600 / 1660, one rank of the 8-way step
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c13"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm or splitk or option_keys" > "$OUT/pytest_gemm.txt" 2>&1; tail -3 "$OUT/pytest_gemm.txt"; grep -E "^FAILED|^E  " "$OUT/pytest_gemm.txt" | head -10
timeout 600 python tools/r05/cp_gemm_ab.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/cp_gemm_ab.txt"
for rep in 1 2; do
  timeout 600 python tools/cp_rank_time.py 8 2>&1 | grep "rank-local"
  VITG_CP_RANK=0 timeout 600 python tools/cp_rank_time.py 8 2>&1 | grep "rank-local"
done | tee "$OUT/cp_rank8.txt"
timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('quick', d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['max_abs_dscore'])" | tee "$OUT/quick.txt"
