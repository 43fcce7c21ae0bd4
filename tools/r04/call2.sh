#!/bin/bash
# round 4, GPU call 2: fused patch-embed loader (tests + in-step), wave-priority A/Bs of the attention kernels (experiment build),
# non-temporal A-operand DMA of the GEMMs (experiment build, in-step A/B)
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c2"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "patch_embed or pair_layout or patchify" > "$OUT/pytest_patch.txt" 2>&1; tail -3 "$OUT/pytest_patch.txt"
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "full_width or golden or dead_row" > "$OUT/pytest_e2e.txt" 2>&1; tail -3 "$OUT/pytest_e2e.txt"
timeout 900 python -m pytest tests/test_fullsize_parity.py -x -q -m gpu -s -k "c2_512x448 and fp16 or b0_512x384 and fp16" > "$OUT/pytest_full.txt" 2>&1; grep -E "^full|passed|failed" "$OUT/pytest_full.txt" | tail -5
EXP="$PWD/videoitg_amd/libvideoitg_hip_exp_nt.so"
for r in 1 2; do
  for bits in 0 1 2 4 8 256 512; do
    echo "## round $r VITG_EXP_ATTN=$bits"
    VITG_LIB_PATH="$EXP" VITG_EXP_ATTN=$bits timeout 300 python tools/attn_bench.py fp16 variant=2 2>&1 | grep -E "vit448 |llm"
  done
done | tee "$OUT/attn_prio_ab.txt"
one() {  # tag, env...
  local tag="$1"; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  python - "$OUT/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o = d.get("roofline_other", {})
print(f"{sys.argv[2]:14s} {d['value']:8.2f} frames/s {d['ms_per_step']:8.2f} ms  GEMM {d['roofline']['achieved']} TF/s  parity {d['parity']['max_abs_dscore']}  "
      f"patchify {o.get('patchify', {}).get('avg_launch_ms')} fused_pe {o.get('patch_embed_fused', {}).get('avg_launch_ms')}")
PY
}
for r in 1 2 3; do
  one "prod_$r" X=1
  one "nt_$r" VITG_LIB_PATH="$EXP"
  one "nt_g8_$r" VITG_LIB_PATH="$EXP" VITG_GEMM_GROUP=8
  one "unfused_$r" VITG_FUSE_PATCHIFY=0
done | tee "$OUT/step_ab.txt"
