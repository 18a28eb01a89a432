set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ZOT_TIMING=2 timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra config5_share_k31 > gpurun_out/c5_t.json 2> gpurun_out/c5_t.err || { tail -5 gpurun_out/c5_t.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/c5_t.json"))
e=d["config5_share_k31"]
print({k:(round(v,1) if isinstance(v,float) else v) for k,v in e.items() if k in ("value","ms_total","cold_ms","verified","table_slab_bytes","unique","error")})
PY
grep -v "^$" gpurun_out/c5_t.err | tail -80
