set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "forced_block_dedupe or large_properties" > gpurun_out/t_dd.log 2>&1 || { tail -30 gpurun_out/t_dd.log; exit 1; }
tail -3 gpurun_out/t_dd.log
for v in 0; do
  ZOT_TUNE=dedupe_variant=$v timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/b_dd$v.json 2> gpurun_out/b_dd$v.err || { tail -5 gpurun_out/b_dd$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/b_dd$v.json"))
print("variant $v", round(d["ms_per_step"],2), d.get("verified_checksums"), d.get("verified_ascending"), {k:round(x["ms_per_step"],2) for k,x in d["pipeline"]["kernels"].items()})
PY
done
export ZOTK_LIB=$PWD/build/libzotk_phases.so
for v in 0; do
  ZOT_TUNE=dedupe_variant=$v timeout -k 10 200 python tools/p0_phases.py 50e6 1 25 > gpurun_out/ph_dd_v$v.json 2> gpurun_out/ph_dd_v$v.err || { tail -5 gpurun_out/ph_dd_v$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ph_dd_v$v.json"))
print("variant $v", d.get("dedupe_cycles_per_block"), d.get("rle_ms"))
PY
done
