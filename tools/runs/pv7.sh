set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_geometry or sort_pairs or finished_in_tiles" > gpurun_out/pv7_test.log 2>&1 || { tail -30 gpurun_out/pv7_test.log; exit 1; }
tail -2 gpurun_out/pv7_test.log
for t in "pairs_variant=2" "pairs_variant=7"; do
  ZOT_TUNE=$t timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra config5_share_k31 > gpurun_out/pv7_$t.json 2> gpurun_out/pv7_$t.err || { tail -5 gpurun_out/pv7_$t.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/pv7_$t.json"))
e=d["config5_share_k31"]
print("$t", {k:(round(v,1) if isinstance(v,float) else v) for k,v in e.items() if k in ("value","ms_total","verified")})
PY
done
