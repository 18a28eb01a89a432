#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. the plain bench line (with extras and CPU baselines)           -> gpurun_out/prof/bench_line.json
#   2. kernel trace + stats of the headline command                  -> gpurun_out/prof/stats/
#   3. two PMC passes (FETCH_SIZE, WRITE_SIZE) of the same command   -> gpurun_out/prof/pmc_f, pmc_w -> pmc_traffic.json
# Counters are collected in their own runs with --kernel-trace only (never with the sys/hip/hsa trace domains).
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify"
timeout -k 10 500 python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err" || { echo "bench failed"; tail -5 "$OUT/bench_line.err"; exit 1; }
echo "bench line ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $CMD > "$OUT/bench_line_under_rocprof.json" 2> "$OUT/stats.err" || { echo "stats run failed"; tail -5 "$OUT/stats.err"; exit 1; }
echo "stats ok"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -- python3 $CMD > /dev/null 2> "$OUT/pmc_f.err" || { echo "pmc fetch failed"; tail -5 "$OUT/pmc_f.err"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -- python3 $CMD > /dev/null 2> "$OUT/pmc_w.err" || { echo "pmc write failed"; tail -5 "$OUT/pmc_w.err"; exit 1; }
echo "pmc ok"
F=$(find "$OUT/pmc_f" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/pmc_w" -name "*counter_collection.csv" | head -1)
KEYS=$(python3 -c "import json; print(json.load(open('$OUT/bench_line.json'))['pipeline']['instances_per_step'] // 2)")
python3 tools/collect_traffic.py "$F" "$W" "$OUT/pmc_traffic.json" "$KEYS" 3 | tail -24
S=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
cp "$S" "$OUT/kernel_stats.csv"
head -20 "$OUT/kernel_stats.csv" | cut -c1-220
# keep what is committed small: the stats summary, the traffic table, the bench lines
rm -rf "$OUT/pmc_f" "$OUT/pmc_w" "$OUT/stats"
ls -la "$OUT"
