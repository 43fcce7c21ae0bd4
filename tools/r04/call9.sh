#!/bin/bash
# round 4, GPU call 9: verification of the tree after the clean-up (experiment hooks removed): whole GPU suite on the product library, the same suite
# on the profiling library, smoke, the driver-style bench line
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c9"; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_gpu.txt" | tail -3
VITG_PROFILING_LIB=1 timeout 3000 python -m pytest tests -q -m gpu > "$OUT/pytest_prof.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_prof.txt" | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$OUT/smoke.txt"
timeout 1500 python bench.py > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"
python - "$OUT/bench_driver_style.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver-style:", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["parity"]["max_abs_dscore"], d["parity"]["pass"], "cpu", d["cpu_baseline"]["value"])
print({k: (v.get("frac"), v.get("avg_launch_ms")) for k, v in d["roofline_other"].items() if isinstance(v, dict)})
print(json.dumps(d["secondary"]["fp8_pareto"]))
print({k: d["secondary"][k].get("value") for k in d["secondary"] if isinstance(d["secondary"][k], dict) and "value" in d["secondary"][k]})
PY
