#!/bin/bash
# round 4, GPU call 28: last tree of the round once more - whole GPU suite on the freshly rebuilt product library, smoke
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c28"; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_gpu.txt" | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$OUT/smoke.txt"
