set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "merge or union or encode or acgt or config4 or config3" > gpurun_out/c4_test.log 2>&1 || { tail -30 gpurun_out/c4_test.log; exit 1; }
tail -2 gpurun_out/c4_test.log
timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra config4_merge_share > gpurun_out/c4.json 2> gpurun_out/c4.err || { tail -5 gpurun_out/c4.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/c4.json"))
e=d["config4_merge_share"]
print(round(e["ms_per_step"],2), e["verified"], {k:round(x["ms_per_step"],2) for k,x in e.get("kernels",{}).items()})
PY
