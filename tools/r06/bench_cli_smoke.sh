mkdir -p gpurun_out/r06_c39
i=0
for args in "--frames 64 --res 224" "--dtype bf16" "--graph" "--videos 2" "--fp8 --fp8-policy llm_mlp" "--config tiny --frames 16" "--frames 520 --res 224"; do
  i=$((i+1))
  timeout 300 python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline $args > gpurun_out/r06_c39/cli_$i.json 2> gpurun_out/r06_c39/cli_$i.err
  rc=$?
  python - "$rc" "$args" gpurun_out/r06_c39/cli_$i.json <<'PY'
import json,sys
rc,args,path=sys.argv[1],sys.argv[2],sys.argv[3]
try:
    line=open(path).read().strip().splitlines()[-1]; d=json.loads(line)
    print(f"rc={rc} [{args}] -> {d['value']} {d['unit']} {d['ms_per_step']} ms, dtype {d['dtype']}, bytes {len(line)}, parity {(d.get('parity') or {}).get('max_abs_dscore')}, workload {d['config']['workload'][:60]}")
except Exception as e:
    print(f"rc={rc} [{args}] -> NO LINE ({type(e).__name__}: {e})")
PY
done
