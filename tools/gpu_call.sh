#!/bin/bash
# ONE parameterised runner for a GPU call (replaces the per-call scripts of rounds 3-5, which are archived as text under tools/archive/):
#     gpurun --timeout 5400 -- 'bash tools/gpu_call.sh <tag> <step> [<step> ...]'
# Everything a step writes goes to gpurun_out/<tag>/ (merged back by gpurun).  Steps:
#   tests[:<pytest -k expression>]   python -m pytest tests -x -q -m gpu            -> pytest_gpu.txt
#   prof-tests                       the same suite against libvideoitg_hip_prof.so (research variants) -> pytest_prof.txt
#   smoke                            __graft_entry__.smoke()                         -> smoke.txt
#   bench[:<extra bench.py flags>]   python bench.py exactly as the driver runs it   -> bench_driver_style.json (+ a one-line digest)
#   quick                            bench.py --no-secondary --no-cpu-baseline --steps 10 -> bench_quick.json
#   profile                          tools/profile_step.sh: rocprofv3 kernel trace + the two PMC traffic passes -> <tag>_kernel_stats_fp16.md, <tag>_pmc_traffic_fp16.md
#   mfma-util                        tools/profile_mfma_util.sh: SQ counters -> MFMA busy per kernel
#   cp[:<world>]                     tools/cp_rank_time.py: one rank's compute of the N-way step (critical rank and rank 0) -> cp_rank<world>.txt
#   py:<script> [args]               any tools/ script, output -> <script>.txt        (quote the whole step)
set -uo pipefail
TAG="${1:?tag}"; shift
OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
digest() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("bench:", d["value"], "frames/s", d["ms_per_step"], "ms | GEMM", r["achieved"], r["frac"], "| parity", (d.get("parity") or {}).get("max_abs_dscore"), (d.get("parity") or {}).get("pass"),
      "| cpu", (d.get("cpu_baseline") or {}).get("value"), "| power", r.get("power_clock") and (r["power_clock"]["power_w"], r["power_clock"]["sclk_ghz"]) or r.get("power_clock_unavailable"))
print("line bytes:", len(open(sys.argv[1]).read().strip().splitlines()[-1]), "| secondary:", d.get("secondary_frames_per_s"))
PY
}
for step in "$@"; do
  name="${step%%:*}"; arg=""; [ "$step" != "$name" ] && arg="${step#*:}"
  case "$name" in
    tests)      if [ -n "$arg" ]; then timeout 3000 python -m pytest tests -x -q -m gpu --durations=40 -k "$arg" > "$OUT/pytest_gpu.txt" 2>&1; else timeout 3000 python -m pytest tests -x -q -m gpu --durations=60 > "$OUT/pytest_gpu.txt" 2>&1; fi
                grep -E "passed|failed|error" "$OUT/pytest_gpu.txt" | tail -3 ;;
    prof-tests) VITG_PROFILING_LIB=1 timeout 3000 python -m pytest tests -q -m gpu > "$OUT/pytest_prof.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_prof.txt" | tail -3 ;;
    smoke)      timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$OUT/smoke.txt" ;;
    bench)      timeout 4500 python bench.py $arg > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"; cp bench_full.json "$OUT/bench_full.json" 2>/dev/null; digest "$OUT/bench_driver_style.json" ;;
    quick)      timeout 900 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 2 > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"; cp bench_full.json "$OUT/bench_quick_full.json" 2>/dev/null; digest "$OUT/bench_quick.json" ;;
    profile)    timeout 1500 bash tools/profile_step.sh "$TAG" fp16 > "$OUT/profile_step.log" 2>&1; tail -3 "$OUT/profile_step.log"; cp gpurun_out/${TAG}_* "$OUT/" 2>/dev/null || true ;;
    mfma-util)  timeout 1500 bash tools/profile_mfma_util.sh "$TAG" fp16 > "$OUT/profile_mfma_util.log" 2>&1; tail -3 "$OUT/profile_mfma_util.log"; cp gpurun_out/${TAG}_* "$OUT/" 2>/dev/null || true ;;
    cp)         W="${arg:-8}"; timeout 600 python tools/cp_rank_time.py "$W" > "$OUT/cp_rank$W.txt" 2>&1; tail -2 "$OUT/cp_rank$W.txt"
                VITG_CP_RANK=0 timeout 600 python tools/cp_rank_time.py "$W" > "$OUT/cp_rank${W}_rank0.txt" 2>&1; tail -2 "$OUT/cp_rank${W}_rank0.txt" ;;
    py)         s="${arg%% *}"; timeout 3000 python tools/$arg > "$OUT/$(basename "$s" .py).txt" 2>&1; tail -15 "$OUT/$(basename "$s" .py).txt" ;;
    *)          echo "unknown step $step"; exit 2 ;;
  esac
done
