#!/bin/bash
# hd-72 attention with a compile-time 16-tile loop (libvideoitg_hip_exp_nkt16.so) against the product kernel, alternating IN ONE CALL:
# the isolated launch (tools/attn_bench.py) and the launch inside the step (bench.py's roofline_other.attention_hd72, HIP events in the side pass)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
OUT=gpurun_out/${1:-r06_attn72}; mkdir -p $OUT
EXP=$PWD/videoitg_amd/libvideoitg_hip_exp_nkt16.so
for r in 1 2 3; do
  for v in product nkt16; do
    if [ $v = nkt16 ]; then export VITG_LIB_PATH=$EXP; else unset VITG_LIB_PATH; fi
    iso=$(python tools/attn_bench.py fp16 variant=2 2>/dev/null | grep -E "^vit448 " | cut -c1-80)
    VITG_BENCH_FULL=$OUT/full_${v}_$r.json python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > $OUT/line_${v}_$r.json 2>/dev/null
    python - $OUT/full_${v}_$r.json $v $r "$iso" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
a = d["roofline_other"]["attention_hd72"]
print(f"round {sys.argv[3]} {sys.argv[2]:8s} step {d['ms_per_step']:.2f} ms ({d['value']:.1f} frames/s)  hd72 in step: {a['avg_launch_ms']:.4f} ms/launch = {a['achieved']:.0f} TF/s ({a['frac']:.3f})  lib {d['library']['path']}  | isolated: {sys.argv[4]}")
PY
  done
done 2>&1 | tee $OUT/ab.txt
