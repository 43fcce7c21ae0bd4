#!/bin/bash
# round 4, GPU call 11: `python bench.py --gpus 8` as a PLAIN command (no launcher) - dry run of the 8-rank job on one GPU (all ranks on cuda:0, gloo);
# timings mean nothing, the JSON line's parity block, rccl_ranks and backend are the point
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c11"; mkdir -p "$OUT"
env -u WORLD_SIZE -u RANK -u LOCAL_RANK VITG_BENCH_SINGLE_DEVICE=1 VITG_DIST_BACKEND=gloo timeout 1200 python bench.py --gpus 8 --steps 2 --warmup 1 > "$OUT/bench_gpus8_dryrun.json" 2> "$OUT/bench_gpus8_dryrun.err"
echo "rc=$?"; tail -c 600 "$OUT/bench_gpus8_dryrun.err"
python - "$OUT/bench_gpus8_dryrun.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(d["n_gpus"], d["rccl_ranks"], d["backend"], d["scaling"], d["value"], d["parity"]["max_abs_dscore"], d["parity"]["pass"], d["config"]["parallelism"])
PY
