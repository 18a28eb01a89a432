# usage: prof_extra.sh <extra name>   -> gpurun_out/prof_<name>_stats.csv (rocprofv3 kernel stats of bench.py --only-extra <name>)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
X=$1
OUT=gpurun_out/prof_$X
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline --only-extra $X > "$OUT/line.json" 2> "$OUT/err.txt" || { tail -5 "$OUT/err.txt"; exit 1; }
S=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
cp "$S" gpurun_out/prof_${X}_stats.csv
rm -rf "$OUT/stats"
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_${X}_stats.csv")))
for r in rows[:22]:
    print("%-64s calls %4s avg %9.3f ms tot %9.2f ms"%(r['Name'][:64],r['Calls'],float(r['AverageNs'])/1e6,float(r['TotalDurationNs'])/1e6))
PY
