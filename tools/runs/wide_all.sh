# The array passes on 8 K-key tiles (two 512-thread workgroups per CU) against 16 K-key tiles (one of 1024 threads), config 2:
# ZK_TUNE_WIDE_TILES 0 / 1 (the default), with the tag pass ranking by LDS adds (tag_words 2) and by ballots (1)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for t in "wide_tiles=0,tag_words=2" "wide_tiles=1,tag_words=2" "wide_tiles=0,tag_words=1" "wide_tiles=1,tag_words=1"; do
  ZOT_TUNE=$t timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/wide_$t.json 2> gpurun_out/wide_$t.err || { tail -5 gpurun_out/wide_$t.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/wide_$t.json"))
print("$t", round(d["ms_per_step"],2), d["verified_checksums"], {k:round(v["ms_per_step"],2) for k,v in d["pipeline"]["kernels"].items() if k in ("pass_keys","pass_packed")})
PY
done
