set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for t in "short_sort=0"; do
  ZOT_TUNE=$t timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra uniform_reads > gpurun_out/uni_$t.json 2> gpurun_out/uni_$t.err || { tail -5 gpurun_out/uni_$t.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/uni_$t.json"))
e=d["uniform_reads"]
print("$t", {k:(round(v,2) if isinstance(v,float) else v) for k,v in e.items() if k in ("value","ms_per_step","verified","unique","error")}, {k:round(x["ms_per_step"],2) for k,x in e.get("kernels",{}).items()})
PY
done
python - <<PY
import json
d=json.load(open("gpurun_out/uni_short_sort=0.json"))
print(d["uniform_reads"].get("reads_40M_in_batches"))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "mirror or large_without or early_collapse or canonical_only" 2>&1 | tail -3
