#!/bin/bash
# round 5, GPU call 5: new parity tests (heavy-tailed fixtures on both tower streams, engine default paths, real-file infer), the 320x256 tile + split-K x3, one rank of the 8-way step
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c5"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or splitk or patch_embed or engine_default or option_keys" > "$OUT/pytest_kernels.txt" 2>&1; tail -4 "$OUT/pytest_kernels.txt"
timeout 1200 python -m pytest tests/test_fullsize_parity.py -x -q -m gpu -s -k "heavy or depth16o or c2o" > "$OUT/pytest_heavy.txt" 2>&1; grep -E "heavy-tailed|full .*o_|passed|failed" "$OUT/pytest_heavy.txt" | tail -12
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "infer" > "$OUT/pytest_infer.txt" 2>&1; tail -2 "$OUT/pytest_infer.txt"
timeout 600 python tools/cp_rank_time.py 8 > "$OUT/cp_rank8.txt" 2>&1; tail -6 "$OUT/cp_rank8.txt"
VITG_CP_RANK=0 timeout 600 python tools/cp_rank_time.py 8 > "$OUT/cp_rank8_rank0.txt" 2>&1; tail -3 "$OUT/cp_rank8_rank0.txt"
timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 5 --warmup 2 > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"; python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('quick', d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['max_abs_dscore'], d['roofline'].get('traffic_over_algorithmic'))" "$OUT/bench_quick.json"
