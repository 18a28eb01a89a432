set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for t in "xcd_group=0" "xcd_group=4" "xcd_group=16"; do
  ZOT_TUNE=$t timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/xg_$t.json 2> gpurun_out/xg_$t.err || { tail -5 gpurun_out/xg_$t.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/xg_$t.json"))
print("$t", round(d["ms_per_step"],2), d["verified_checksums"], {k:round(v["ms_per_step"],2) for k,v in d["pipeline"]["kernels"].items() if k in ("pass_keys","pass_packed","pass_stream")})
PY
done
