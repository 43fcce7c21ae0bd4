#!/bin/bash
# round 3, GPU call 29: the fp8 policy sweeps (plain and planted case) and the two fp8 job lines on the final tree
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c29"; mkdir -p "$OUT"
timeout 1800 python tools/fp8_policy_sweep.py c2_512x448 2>&1 | grep -v amdgpu.ids | tee "$OUT/fp8_policy_sweep.txt"
timeout 1800 python tools/fp8_policy_sweep.py c2p_512x448 2>&1 | grep -v amdgpu.ids | tee "$OUT/fp8_policy_sweep_planted.txt"
timeout 900 python bench.py --fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_fp8_job_all.json" 2> "$OUT/bench_fp8_job_all.err"
timeout 900 python bench.py --fp8 --fp8-policy llm_mlp --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_fp8_job_llm_mlp.json" 2> "$OUT/bench_fp8_job_llm_mlp.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("all", "llm_mlp"):
    d = json.loads(open(f"{sys.argv[1]}/bench_fp8_job_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["parity"]["max_abs_dscore"], d["parity"]["pass"])
PY
