#!/bin/bash
# round 3, GPU call 9: final state - whole GPU suite, smoke, every bench mode, the default bench line, rocprofv3 summaries of the same command,
# one rank of the context-parallel step with and without the K|V-first overlap split
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c9"; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1; tail -4 "$OUT/pytest_gpu.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --graph --no-cpu-baseline --no-secondary > "$OUT/bench_graph.json" 2> "$OUT/bench_graph.err"; tail -c 400 "$OUT/bench_graph.json"; echo
timeout 900 python bench.py --videos 2 --steps 1 --warmup 1 > "$OUT/bench_videos2.json" 2> "$OUT/bench_videos2.err"; tail -c 600 "$OUT/bench_videos2.json"; echo
timeout 900 python bench.py --fp8 --no-cpu-baseline --no-secondary > "$OUT/bench_fp8_all.json" 2> "$OUT/bench_fp8_all.err"; python -c "
import json; r=json.loads(open('$OUT/bench_fp8_all.json').read().strip().splitlines()[-1]); print('fp8 all:', r['value'], r['dtype'][:60], r['parity']['max_abs_dscore'], r['parity']['pass'], r['roofline']['frac'])"
timeout 900 python bench.py --fp8 --fp8-policy llm_mlp --no-cpu-baseline --no-secondary > "$OUT/bench_fp8_llm_mlp.json" 2> "$OUT/bench_fp8_llm_mlp.err"; python -c "
import json; r=json.loads(open('$OUT/bench_fp8_llm_mlp.json').read().strip().splitlines()[-1]); print('fp8 llm_mlp:', r['value'], r['dtype'][:60], r['parity']['max_abs_dscore'], r['parity']['pass'], r['parity']['top32_set_match'])"
for w in 8; do for ov in 1 0; do echo "VITG_CP_OVERLAP=$ov"; VITG_CP_OVERLAP=$ov timeout 600 python tools/cp_rank_time.py $w 2>&1 | tail -2; done; done | tee "$OUT/cp_rank_time.txt"
timeout 1500 bash tools/profile_step.sh r03 fp16 > "$OUT/profile_step.log" 2>&1; tail -3 "$OUT/profile_step.log"
cp gpurun_out/r03_kernel_stats_fp16.md gpurun_out/r03_pmc_traffic_fp16.md gpurun_out/r03_gemm_traffic_fp16.json gpurun_out/r03_prof_bench_fp16.json "$OUT/" 2>/dev/null
timeout 1500 python bench.py --steps 5 --warmup 2 > "$OUT/bench_full_default.json" 2> "$OUT/bench_full_default.err"; python -c "
import json; r=json.loads(open('$OUT/bench_full_default.json').read().strip().splitlines()[-1]); print('default:', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['roofline']['traffic'], r['parity']['max_abs_dscore'], r['parity']['pass'], r['cpu_baseline']['value'], r['cpu_baseline']['cores'], {k: (v.get('value'), (v.get('parity') or v.get('parity_vs_oracle') or {}).get('max_abs_dscore')) for k, v in r['secondary'].items()})"
