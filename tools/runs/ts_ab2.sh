set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kmerize or mirror or large_without or early_collapse or canonical_only" > gpurun_out/ts_test2.log 2>&1 || { tail -30 gpurun_out/ts_test2.log; exit 1; }
tail -3 gpurun_out/ts_test2.log
for t in "tile_sort=1"; do
  ZOT_TUNE=$t timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra config5_share_k31 > gpurun_out/tsc5_$t.json 2> gpurun_out/tsc5_$t.err || { tail -5 gpurun_out/tsc5_$t.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/tsc5_$t.json"))
e=d["config5_share_k31"]
print("$t", {k:(round(v,1) if isinstance(v,float) else v) for k,v in e.items() if k not in ("workload","kernels","roofline")})
PY
done
