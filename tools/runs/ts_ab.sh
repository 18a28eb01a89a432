set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for t in "tile_sort=0" "tile_sort=1"; do
  ZOT_TUNE=$t timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra uniform_reads > gpurun_out/tsu_$t.json 2> gpurun_out/tsu_$t.err || { tail -5 gpurun_out/tsu_$t.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/tsu_$t.json"))
e=d["uniform_reads"]
print("$t", {k:(round(v,2) if isinstance(v,float) else v) for k,v in e.items() if k in ("value","ms_per_step","verified","unique","error")}, {k:(x["launches"], round(x["ms_per_step"],2)) for k,x in e.get("kernels",{}).items()})
print(e.get("reads_40M_in_batches"))
PY
done
for t in "tile_sort=0" "tile_sort=1"; do
  ZOT_TUNE=$t timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra config5_share_k31 > gpurun_out/tsc5_$t.json 2> gpurun_out/tsc5_$t.err || { tail -5 gpurun_out/tsc5_$t.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/tsc5_$t.json"))
e=d["config5_share_k31"]
print("$t", {k:(round(v,1) if isinstance(v,float) else v) for k,v in e.items() if k in ("value","ms_total","cold_ms","verified","table_slab_bytes","unique","error")})
print({k:(x["launches"], round(x.get("ms_per_step", x.get("ms",0)),2)) for k,x in e.get("kernels",{}).items()})
PY
done
