#!/bin/bash
# round 3, GPU call 37: hd-128 attention with the scale / max-subtract FMA pinned to v_pk_fma_f32 (attn_sp=4 = the compiler's two v_fma_f32 per pair); VALU issue rates
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c37"; mkdir -p "$OUT"
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate valu_rate.hip && /tmp/valu_rate) 2>&1 | tail -8 | tee "$OUT/valu_rate.txt"
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "attention and 128" -x -q 2>&1 | tail -2
for r in 1 2 3; do
  echo "## round $r: v_pk_fma_f32 (default)"; timeout 300 python tools/attn_bench.py fp16 2>&1 | grep -v amdgpu.ids | tail -1
  echo "## round $r: compiler's form (attn_sp=4)"; timeout 300 python tools/attn_bench.py fp16 attn_sp=4 2>&1 | grep -v amdgpu.ids | tail -1
done | tee "$OUT/attn_sp.txt"
for r in 1 2; do for sp in 0 4; do
  VITG_ATTN_SP=$sp timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/b.json" 2> "$OUT/b.err"
  python - "$OUT/b.json" $sp <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d['roofline_other']['attention_hd128']
print(f"attn_sp={sys.argv[2]}: {d['value']} frames/s {d['ms_per_step']} ms  hd128 {h['achieved']} TF/s frac {h['frac']} avg {h['avg_launch_ms']} ms parity {d['parity']['max_abs_dscore']}")
PY
done; done | tee "$OUT/step_ab.txt"
