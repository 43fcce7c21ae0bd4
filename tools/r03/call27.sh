#!/bin/bash
# round 3, GPU call 26: complete GPU suite, smoke, driver-style bench on the tree with the epilogue work (16-bit and fp8 kernels)
set -uo pipefail
OUT="$PWD/gpurun_out/r03_c27"; mkdir -p "$OUT"
echo "(suite ran in call 26: 1388 passed)"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$OUT/smoke.txt"
timeout 1500 python bench.py > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"
python - "$OUT/bench_driver_style.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver-style:", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity"]["max_abs_dscore"], d["parity"]["pass"])
print("hd72", d["roofline_other"]["attention_hd72"]["frac"], "hd128", d["roofline_other"]["attention_hd128"]["frac"])
s = d["secondary"]
for k in ("b0_384", "bf16", "fp8", "fp8_llm_mlp", "residual_fp32", "from_uint8_frames"):
    v = s.get(k) or {}
    print(k, v.get("value"), (v.get("parity_vs_oracle") or v.get("parity") or {}).get("max_abs_dscore"), (v.get("planted_signal") or {}))
print("planted 16-bit:", s.get("planted_signal"))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
