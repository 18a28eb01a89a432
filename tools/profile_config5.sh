#!/bin/bash
# rocprofv3 evidence for one GPU's share of config 5 (K = 31): kernel stats + two PMC passes (FETCH_SIZE, WRITE_SIZE), as tools/profile_round.sh
# does for the headline.  Run through gpurun from the repo root; results under gpurun_out/prof5/.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof5
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="bench.py --only-extra config5_share_k31"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $CMD > "$OUT/line.json" 2> "$OUT/stats.err" || { echo "stats failed"; tail -3 "$OUT/stats.err"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -- python3 $CMD > /dev/null 2> "$OUT/pmc_f.err" || { echo "pmc fetch failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -- python3 $CMD > /dev/null 2> "$OUT/pmc_w.err" || { echo "pmc write failed"; exit 1; }
F=$(find "$OUT/pmc_f" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/pmc_w" -name "*counter_collection.csv" | head -1)
python3 tools/collect_traffic.py "$F" "$W" "$OUT/pmc_traffic.json" 4430729197 2 | tail -20
cp "$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/pmc_f" "$OUT/pmc_w" "$OUT/stats"
