#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + the two HBM-traffic PMC passes of the default bench workload.
#   gpurun --timeout 1500 -- 'bash tools/profile_step.sh r02 fp16'
# Writes gpurun_out/<tag>_kernel_stats_<dtype>.md, gpurun_out/<tag>_pmc_traffic_<dtype>.md and refreshes profiles/gemm_traffic.json in
# the scratch copy (copy the printed JSON back by hand: only gpurun_out/ is merged).  Counters are collected in their own runs,
# without any trace domain (the pool refuses --pmc combined with sys/hip/hsa traces).
set -uo pipefail
TAG="${1:-r02}"; DT="${2:-fp16}"
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 3 --warmup 1 --dtype $DT --no-cpu-baseline --no-secondary"
cd /tmp
REPO="$OLDPWD"
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o bench -- $CMD > "$OUT/${TAG}_prof_bench_${DT}.json" 2> "$OUT/${TAG}_prof_kt.err"
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python "$OLDPWD/tools/rocpd_summary.py" "$DB" > "$OUT/${TAG}_kernel_stats_${DT}.md" 2>> "$OUT/${TAG}_prof_kt.err"
# counters only on this library's kernels (the torch integer kernels that generate the synthetic weights are not of interest, and
# profiling them crashed rocprofv3's counter service on this image); one timed step is enough for per-launch averages
PCMD="python $REPO/bench.py --steps 1 --warmup 0 --dtype $DT --no-cpu-baseline --no-secondary"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "vitg" -d /tmp/prof_f -o bench -- $PCMD > /dev/null 2> "$OUT/${TAG}_prof_f.err"
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "vitg" -d /tmp/prof_w -o bench -- $PCMD > /dev/null 2> "$OUT/${TAG}_prof_w.err"
F=$(find /tmp/prof_f -name "*.db" | head -1); W=$(find /tmp/prof_w -name "*.db" | head -1)
cd "$OLDPWD"
python tools/rocpd_pmc.py "$F" "$W" > "$OUT/${TAG}_pmc_traffic_${DT}.md" 2>> "$OUT/${TAG}_prof_f.err"
cp profiles/gemm_traffic.json "$OUT/${TAG}_gemm_traffic_${DT}.json"
tail -3 "$OUT/${TAG}_kernel_stats_${DT}.md"; tail -2 "$OUT/${TAG}_pmc_traffic_${DT}.md"
