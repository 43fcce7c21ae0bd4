#!/bin/bash
# round 4, GPU call 33: fp8 policy "seven" (all layer GEMMs but the tower's out-proj) against "all" as the job, three alternations
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c33"; mkdir -p "$OUT"
one() {
  local tag="$1" pol="$2"
  timeout 600 python bench.py --fp8 --fp8-policy "$pol" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:10s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  fp8 GEMM {r['achieved']} TF/s = {r['frac']} ({r['launches']} launches)  max dscore {d['parity']['max_abs_dscore']}  top32 kept {d['parity']['topk_ladder']['top32']['kept']}  top1 {d['parity']['top1_kept']}")
PY
}
for r in 1 2 3; do
  one "all_$r" all
  one "seven_$r" seven
done | tee "$OUT/fp8_policy_ab.txt"
