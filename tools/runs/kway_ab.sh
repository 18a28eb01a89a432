set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multigpu.py tests/test_gpu_exchange.py tests/test_gpu_commands.py -x -q -k "merge" > gpurun_out/t_kw.log 2>&1 || { tail -40 gpurun_out/t_kw.log; exit 1; }
tail -3 gpurun_out/t_kw.log
for v in 1 0; do
  ZOT_TUNE=kway=$v timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra config4_merge_share > gpurun_out/c4_$v.json 2> gpurun_out/c4_$v.err || { tail -5 gpurun_out/c4_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/c4_$v.json"))["config4_merge_share"]
print("kway $v", {k:(round(x,3) if isinstance(x,float) else x) for k,x in d.items() if k in ("value","ms_per_step","verified","pairs_in","pairs_out")}, {k:round(x["ms_per_step"],2) for k,x in d.get("kernels",{}).items()})
PY
done
