#!/bin/bash
# round 5, GPU call 8: whole GPU suite, smoke, driver-style bench (all secondaries + cpu baseline), rocprofv3 kernel trace + PMC traffic + MFMA utilisation
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c8"; mkdir -p "$OUT"
timeout 3000 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_gpu.txt" | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$OUT/smoke.txt"
timeout 2400 python bench.py > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"
python - "$OUT/bench_driver_style.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("driver-style:", d["value"], d["ms_per_step"], r["achieved"], r["frac"], d["parity"]["max_abs_dscore"], d["parity"]["pass"], "cpu", d["cpu_baseline"]["value"])
print("traffic", r.get("traffic"), r.get("traffic_algorithmic"), r.get("traffic_over_algorithmic"), "power", r.get("power_clock"), r.get("power_clock_unavailable"))
print({k: d["secondary"][k].get("value") for k in d["secondary"] if isinstance(d["secondary"][k], dict) and "value" in d["secondary"][k]})
print("heavy_tailed", d["secondary"].get("heavy_tailed"))
print({k: (v.get("achieved"), v.get("frac"), v.get("share_of_step")) for k, v in d.get("roofline_other", {}).items() if isinstance(v, dict)})
PY
timeout 1500 bash tools/profile_step.sh r05 fp16 > "$OUT/profile_step.log" 2>&1; tail -3 "$OUT/profile_step.log"
cp gpurun_out/r05_* "$OUT/" 2>/dev/null || true
