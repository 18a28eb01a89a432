set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kmerize or mirror or large_without or early_collapse or canonical_only or sort" > gpurun_out/ts_test2.log 2>&1 || { tail -30 gpurun_out/ts_test2.log; exit 1; }
tail -3 gpurun_out/ts_test2.log
for x in uniform_reads config5_share_k31; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra $x > gpurun_out/ts3_$x.json 2> gpurun_out/ts3_$x.err || { tail -5 gpurun_out/ts3_$x.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ts3_$x.json"))
e=d["$x"]
print("$x", {k:(round(v,1) if isinstance(v,float) else v) for k,v in e.items() if k not in ("workload","kernels","roofline","verified_by","reads_40M_in_batches")})
print({k:(x["launches"], round(x.get("ms_per_step", x.get("ms",0)),2)) for k,x in e.get("kernels",{}).items()})
print(e.get("reads_40M_in_batches"))
PY
done
