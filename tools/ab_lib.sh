#!/bin/bash
# same-box A/B of two builds of the product library: videoitg_amd/libvideoitg_hip.so (new) against videoitg_amd/libvideoitg_hip_old.so
# (build HEAD into /tmp and copy it there); per shape with the real epilogues, then the step, two alternations
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
L=videoitg_amd
cp $L/libvideoitg_hip.so /tmp/new.so; cp $L/libvideoitg_hip_old.so /tmp/old.so
for r in 1 2; do
  for v in new old; do
    cp /tmp/$v.so $L/libvideoitg_hip.so
    echo "#### $v run $r"; python tools/gemm_cfg_sweep.py fp16 2>&1 | grep -E "vit_fc1|vit_qkv" | cut -c1-80
    python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['parity']['max_abs_dscore'])"
  done
done
cp /tmp/new.so $L/libvideoitg_hip.so
