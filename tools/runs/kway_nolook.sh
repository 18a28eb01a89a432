set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in 0 1; do
  ZK_KWAY_NOLOOK=$v ZOTK_LIB=build/libzotk_phases.so timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra config4_merge_share > gpurun_out/kwn_$v.json 2> gpurun_out/kwn_$v.err || { tail -5 gpurun_out/kwn_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/kwn_$v.json"))
e=d["config4_merge_share"]
print("nolook=$v", round(e["ms_per_step"],2), e["verified"], {k:round(x["ms_per_step"],2) for k,x in e.get("kernels",{}).items()})
PY
done
