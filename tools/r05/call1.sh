#!/bin/bash
# round 5, GPU call 1: C4 at its own workload (8 videos x 512 x 448^2 through ONE captured graph), box baseline (quick bench without secondaries)
set -uo pipefail
OUT="$PWD/gpurun_out/r05_c1"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_graphs.py -x -q -m gpu -s -k "batch_of_eight or full_size_graph" > "$OUT/pytest_c4.txt" 2>&1; tail -5 "$OUT/pytest_c4.txt"
timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 2 > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"
python - "$OUT/bench_quick.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("quick:", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["parity"]["max_abs_dscore"], d["parity"]["pass"], d["roofline"].get("power_clock"))
PY
