set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "forced_block_dedupe or large_properties or stream_ranges or large_without" > gpurun_out/t_tp.log 2>&1 || { tail -40 gpurun_out/t_tp.log; exit 1; }
tail -3 gpurun_out/t_tp.log
for v in 0 1; do
  ZOT_TUNE=tag_pass=$v timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/b_tp$v.json 2> gpurun_out/b_tp$v.err || { tail -5 gpurun_out/b_tp$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/b_tp$v.json"))
print("tag_pass $v", round(d["ms_per_step"],2), d.get("verified_checksums"), d.get("verified_ascending"), {k:round(x["ms_per_step"],2) for k,x in d["pipeline"]["kernels"].items()})
PY
done
