cd $GRAFT_REPO_ROOT
L=videoitg_amd
cp $L/libvideoitg_hip.so /tmp/new.so; cp $L/libvideoitg_hip_old.so /tmp/old.so
for r in 1 2; do
  for v in new old; do
    cp /tmp/$v.so $L/libvideoitg_hip.so
    echo "#### $v run $r"; python tools/gemm_fp8_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-112
  done
done > gpurun_out/r02_epi_ab.log 2>&1
for r in 1 2; do
  for v in new old; do
    cp /tmp/$v.so $L/libvideoitg_hip.so
    echo "#### $v run $r"; python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['achieved'], r['parity']['max_abs_dscore'])"
  done
done > gpurun_out/r02_epi_ab_step.log 2>&1
cp /tmp/new.so $L/libvideoitg_hip.so
VITG_PROFILING_LIB=1 python tools/gemm_timeline.py fp16 vit_qkv,vit_fc1,vit_out,vit_fc2,llm_qkv,llm_gu 2>&1 | grep -v "tile 0\|tile 2\|amdgpu.ids" > gpurun_out/r02_timeline4.log
