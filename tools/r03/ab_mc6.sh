#!/bin/bash
# same-box A/B: videoitg_amd/libvideoitg_hip.so (MC = 4) against libvideoitg_hip_mc6.so (192-row wave tiles in intervals of 6 activation tiles)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
L=videoitg_amd
cp $L/libvideoitg_hip.so /tmp/base.so; cp $L/libvideoitg_hip_mc6.so /tmp/mc6.so
for r in 1 2; do
  for v in base mc6; do
    cp /tmp/$v.so $L/libvideoitg_hip.so
    echo "#### $v run $r"; python tools/gemm_cfg_sweep.py fp16 2>&1 | grep -E "vit_fc1|llm_gu|llm_o " | cut -c1-120
    python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['parity']['max_abs_dscore'])"
  done
done
cp /tmp/base.so $L/libvideoitg_hip.so
