#!/bin/bash
# round 4, GPU call 30: tree with the fp8 tower on the 16-bit stream - whole GPU suite (product), profiling-library files, fp8 job A/B with the
# res16 launches inside the fp8 roofline, driver-style bench
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c30"; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_gpu.txt" | tail -3
VITG_PROFILING_LIB=1 timeout 1800 python -m pytest tests/test_gpu_profiling_lib.py tests/test_gpu_fp8.py tests/test_gpu_e2e.py -q -m gpu > "$OUT/pytest_prof.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_prof.txt" | tail -3
one() {
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --fp8 --fp8-policy all --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:12s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  fp8 GEMM {r['achieved']} TF/s = {r['frac']} ({r['launches']} launches, {r['avg_launch_ms']} ms)  max dscore {d['parity']['max_abs_dscore']}  top32 kept {d['parity']['topk_ladder']['top32']['kept']}")
PY
}
for r in 1 2; do
  one "fp32stream_$r" VITG_FP8_R16=0
  one "stream16_$r" VITG_FP8_R16=1
done | tee "$OUT/fp8_job_ab.txt"
timeout 1500 python bench.py > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"
python - "$OUT/bench_driver_style.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver-style:", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["parity"]["max_abs_dscore"], d["parity"]["pass"], "cpu", d["cpu_baseline"]["value"])
for k in ("fp8_llm_mlp", "fp8_llm", "fp8"):
    s = d["secondary"][k]
    print(k, s["value"], s["roofline"]["achieved"], s["roofline"]["frac"], s["roofline"]["launches"], s["parity_vs_oracle"]["max_abs_dscore"], s["parity_vs_oracle"]["topk_ladder"]["top32"]["kept"])
print(json.dumps(d["roofline"].get("power_clock")))
PY
