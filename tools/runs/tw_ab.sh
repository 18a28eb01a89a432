set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forced_block or sort_keys" > gpurun_out/tw_test.log 2>&1 || { tail -30 gpurun_out/tw_test.log; exit 1; }
tail -2 gpurun_out/tw_test.log
for t in "tag_words=1" "tag_words=2" "tag_words=1" "tag_words=2"; do
  ZOT_TUNE=$t timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/tw_$t.json 2> gpurun_out/tw_$t.err || { tail -5 gpurun_out/tw_$t.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/tw_$t.json"))
print("$t", round(d["ms_per_step"],2), d["verified_checksums"], d["verified_ascending"], {k:round(v["ms_per_step"],2) for k,v in d["pipeline"]["kernels"].items() if k in ("pass_keys","pass_stream","rle")})
PY
done
