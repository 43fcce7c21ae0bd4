#!/bin/bash
# round 4, GPU call 35: LAST tree of the round (fp8 tower on the 16-bit stream, fp8 policy seven in the line) - whole GPU suite, profiling-library files, smoke, driver-style bench, rocprofv3 trace + PMC
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c35"; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_gpu.txt" | tail -3
VITG_PROFILING_LIB=1 timeout 1800 python -m pytest tests/test_gpu_profiling_lib.py tests/test_gpu_fp8.py tests/test_gpu_e2e.py -q -m gpu > "$OUT/pytest_prof.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_prof.txt" | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$OUT/smoke.txt"
timeout 1500 python bench.py > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"
python - "$OUT/bench_driver_style.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver-style:", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["parity"]["max_abs_dscore"], d["parity"]["pass"], "cpu", d["cpu_baseline"]["value"])
print({k: d["secondary"][k].get("value") for k in d["secondary"] if isinstance(d["secondary"][k], dict) and "value" in d["secondary"][k]})
PY
timeout 1500 bash tools/profile_step.sh r04d fp16 > "$OUT/profile_step.log" 2>&1; tail -3 "$OUT/profile_step.log"
cp gpurun_out/r04d_* "$OUT/" 2>/dev/null || true
