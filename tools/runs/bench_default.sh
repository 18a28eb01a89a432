set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
SECONDS=0; timeout -k 10 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -5 gpurun_out/bench_default.err; exit 1; }
echo "elapsed $SECONDS s"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["verified_checksums"], d["verified_ascending"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic"))
for k,v in d.get("extra",{}).items():
    if isinstance(v,dict): print(k, v.get("value"), v.get("ms_per_step", v.get("ms_total")), v.get("verified"), v.get("error"))
print(d.get("cpu_baseline",{}).get("value"))
PY
