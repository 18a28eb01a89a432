set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for t in "tag_words=2" "tag_words=1" "tag_words=0"; do
  ZOT_TUNE=$t timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/tw2_$t.json 2> gpurun_out/tw2_$t.err || { tail -5 gpurun_out/tw2_$t.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/tw2_$t.json"))
print("$t", round(d["ms_per_step"],2), d["verified_checksums"], {k:round(v["ms_per_step"],2) for k,v in d["pipeline"]["kernels"].items() if k in ("pass_keys","rle","pass_stream")})
PY
done
