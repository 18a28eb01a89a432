set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for t in "pairs_variant=2" "pairs_variant=4" "pairs_variant=6"; do
  ZOT_TUNE=$t timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra config5_share_k31 > gpurun_out/pv_$t.json 2> gpurun_out/pv_$t.err || { tail -5 gpurun_out/pv_$t.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/pv_$t.json"))
e=d["config5_share_k31"]
print("$t", {k:(round(v,1) if isinstance(v,float) else v) for k,v in e.items() if k in ("value","ms_total","verified")})
PY
done
