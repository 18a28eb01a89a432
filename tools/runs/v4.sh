set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1 || { tail -30 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
for x in uniform_reads config5_share_k31 config2_variable_length; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra $x > gpurun_out/v4_$x.json 2> gpurun_out/v4_$x.err || { tail -5 gpurun_out/v4_$x.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/v4_$x.json"))
e=d["$x"]
print("$x", {k:(round(v,1) if isinstance(v,float) else v) for k,v in e.items() if k in ("value","ms_per_step","ms_total","verified")}, {k:(x["launches"], round(x.get("ms_per_step",0),2)) for k,x in e.get("kernels",{}).items() if k in ("pass_keys","tile_sort")})
PY
done
