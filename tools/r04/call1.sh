#!/bin/bash
# round 4, GPU call 1: the new tests (C3 at its own workload: 8 / 4 / 2 gloo ranks on one GPU, full model; bench.py spawning its own
# ranks; multi-visual plugin documents), the per-stage error profile of chunk520, the driver-style bench line on this box
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c1"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_cp_fullsize.py tests/test_lmms_plugin.py -x -q -m gpu -s > "$OUT/pytest_new.txt" 2>&1
grep -E "context-parallel x|passed|failed|Error|error" "$OUT/pytest_new.txt" | tail -20
timeout 600 python tools/stage_diff.py chunk520 > "$OUT/stage_diff_chunk520.txt" 2>&1; tail -25 "$OUT/stage_diff_chunk520.txt"
timeout 1500 python bench.py > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"
python - "$OUT/bench_driver_style.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver-style:", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["parity"]["max_abs_dscore"], d["parity"]["pass"], "cpu", d["cpu_baseline"]["value"])
for k in ("fp8_llm_mlp", "fp8_llm", "fp8"):
    s = d["secondary"][k]
    print(k, s["value"], s["parity_vs_oracle"]["max_abs_dscore"], "same frames", s["selects_same_frames"], "top1", s["top1_kept"], s["roofline"]["frac"])
print(json.dumps(d["secondary"]["fp8_pareto"]))
print({k: (v.get("frac"), v.get("avg_launch_ms")) for k, v in d["roofline_other"].items() if isinstance(v, dict)})
PY
