#!/bin/bash
# round 3, GPU call 12: the tree as committed - whole GPU suite, smoke, the default bench line (driver-style K / W) and the rocprofv3 summaries of the same command
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c12"; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 bash tools/profile_step.sh r03 fp16 > "$OUT/profile_step.log" 2>&1; tail -2 "$OUT/profile_step.log"
cp gpurun_out/r03_kernel_stats_fp16.md gpurun_out/r03_pmc_traffic_fp16.md gpurun_out/r03_gemm_traffic_fp16.json gpurun_out/r03_prof_bench_fp16.json "$OUT/" 2>/dev/null
cp gpurun_out/r03_gemm_traffic_fp16.json profiles/gemm_traffic.json 2>/dev/null
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"; python -c "
import json; r=json.loads(open('$OUT/bench_driver_style.json').read().strip().splitlines()[-1]); print('driver-style:', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['roofline']['traffic'], r['parity']['max_abs_dscore'], r['parity']['pass'], r['cpu_baseline']['value'], r['cpu_baseline']['cores'], {k: (v.get('value'), (v.get('parity') or v.get('parity_vs_oracle') or {}).get('max_abs_dscore')) for k, v in r['secondary'].items()}); print({k:(v.get('achieved'), v.get('frac')) for k,v in r['roofline_other'].items() if isinstance(v,dict)})"
for w in 8; do timeout 600 python tools/cp_rank_time.py $w 2>&1 | tail -2 | head -1; done | tee "$OUT/cp_rank_time.txt"
