#!/bin/bash
# round 3, GPU call 30: what would a bias-free residual GEMM be worth (timing experiment, wrong results)
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c30"; mkdir -p "$OUT"
for r in 1 2 3; do
  for nb in 0 1; do
    VITG_EXP_NOBIAS=$nb timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_nb${nb}_$r.json" 2> "$OUT/bench_nb${nb}_$r.err"
    python - "$OUT/bench_nb${nb}_$r.json" $nb <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"nobias={sys.argv[2]}: {d['value']} frames/s {d['ms_per_step']} ms  GEMM {d['roofline']['achieved']} TF/s")
PY
  done
done | tee "$OUT/step_ab.txt"
timeout 900 python bench.py --fp8 --fp8-policy llm_mlp --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_fp8_job_llm_mlp.json" 2> "$OUT/bench_fp8_job_llm_mlp.err"
python - "$OUT/bench_fp8_job_llm_mlp.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fp8 job llm_mlp:", d["value"], d["ms_per_step"], d["roofline"])
PY
