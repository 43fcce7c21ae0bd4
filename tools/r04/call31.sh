#!/bin/bash
# round 4, GPU call 31: `python bench.py --gpus 8` typed as a plain command on the last tree (one-GPU dry run: 8 self-spawned ranks share cuda:0 over gloo;
# the timing means nothing, the parity block and rccl_ranks do) + one rank of the 8 / 4 / 2-way step
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c31"; mkdir -p "$OUT"
VITG_BENCH_SINGLE_DEVICE=1 VITG_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/bench_gpus8_dryrun.json" 2> "$OUT/bench_gpus8_dryrun.err"
python - "$OUT/bench_gpus8_dryrun.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("gpus 8 dry run:", d["n_gpus"], d["rccl_ranks"], d["backend"], d["value"], d["parity"]["max_abs_dscore"], d["parity"]["pass"], d["parity"]["top32_set_match"])
PY
for w in 8 4 2; do timeout 600 python tools/cp_rank_time.py $w 2>&1 | grep "world="; done | tee "$OUT/cp_rank_time.txt"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one GPU on this box:', d['value'], d['ms_per_step'])" | tee -a "$OUT/cp_rank_time.txt"
