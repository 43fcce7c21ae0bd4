#!/bin/bash
# round 4, GPU call 25: bench.py with roofline.power_clock exactly as the driver runs it + the tests that run bench.py
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c25"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "bench" 2>&1 | tail -2 | tee "$OUT/pytest_bench.txt"
timeout 1500 python bench.py > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"
python - "$OUT/bench_driver_style.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver-style:", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["parity"]["max_abs_dscore"], d["parity"]["pass"], "cpu", d["cpu_baseline"]["value"])
print(json.dumps(d["roofline"].get("power_clock")))
print({k: d["secondary"][k].get("value") for k in d["secondary"] if isinstance(d["secondary"][k], dict) and "value" in d["secondary"][k]})
print({k: (v.get("achieved"), v.get("frac")) for k, v in d["roofline_other"].items() if isinstance(v, dict)})
PY
