#!/bin/bash
# round 3, GPU call 34: interior forms in the shared staged epilogue (fp8 kernel, 32x32 cross-check kernels): tests and the fp8 tiers
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c34"; mkdir -p "$OUT"
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp8.py -x -q > "$OUT/pytest.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest.txt" | tail -2
VITG_PROFILING_LIB=1 timeout 1800 python -m pytest tests/test_gpu_profiling_lib.py -q > "$OUT/pytest_prof.txt" 2>&1; tail -1 "$OUT/pytest_prof.txt"
for r in 1 2; do
timeout 900 python bench.py --fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_fp8_job_all.json" 2> "$OUT/bench_fp8_job_all.err"
timeout 900 python bench.py --fp8 --fp8-policy llm_mlp --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_fp8_job_llm_mlp.json" 2> "$OUT/bench_fp8_job_llm_mlp.err"
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_16.json" 2> "$OUT/bench_16.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("fp8_job_all", "fp8_job_llm_mlp", "16"):
    d = json.loads(open(f"{sys.argv[1]}/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["parity"]["max_abs_dscore"], d["parity"]["pass"])
PY
done
