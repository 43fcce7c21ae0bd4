#!/bin/bash
# round 4, GPU call 19: (a) hd-72 attention with K/V in a head-major layout (strides only, same kernel): is the L2 -> LDS stream of the packed rows
# (144 B of every 256 B fetched) worth a scatter epilogue in the qkv GEMM?  (b) the LayerNorm tests on the product library with the four-rows kernel
set -uo pipefail
OUT="$PWD/gpurun_out/r04_c19"; mkdir -p "$OUT"
timeout 600 python tools/experiments/attn72_layout_ab.py 512 2>&1 | grep -v amdgpu.ids | tee "$OUT/attn72_layout_ab.txt"
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm" 2>&1 | tail -2 | tee "$OUT/pytest_ln.txt"
