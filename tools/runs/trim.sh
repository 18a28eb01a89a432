set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra trim > gpurun_out/trim.json 2> gpurun_out/trim.err || { tail -5 gpurun_out/trim.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/trim.json"))
e=d["trim"]
print({k:v for k,v in e.items() if k not in ("workload","verified_by")})
PY
