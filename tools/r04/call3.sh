#!/bin/bash
# round 4, GPU call 3: the whole GPU suite on the slimmed product library (four option keys), the research cases on the profiling library
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c3"; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_gpu.txt" | tail -3
VITG_PROFILING_LIB=1 timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_profiling_lib.py -q -m gpu > "$OUT/pytest_prof.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_prof.txt" | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$OUT/smoke.txt"
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench:", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["parity"]["max_abs_dscore"], d["parity"]["pass"])
PY
