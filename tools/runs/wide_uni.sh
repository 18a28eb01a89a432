set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in "" "1"; do
  env ${v:+ZK_WIDE_ALL=1} ZOTK_LIB=build/libzotk_phases.so timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra uniform_reads > gpurun_out/wu_$v.json 2> gpurun_out/wu_$v.err || { tail -5 gpurun_out/wu_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/wu_$v.json"))
e=d["uniform_reads"]
print("wide_all='$v'", round(e["ms_per_step"],2), e["verified"], {k:(x["launches"], round(x["ms_per_step"],2)) for k,x in e.get("kernels",{}).items() if k in ("pass_keys","tile_sort","pass_stream")})
PY
done
