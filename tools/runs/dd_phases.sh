set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export ZOTK_LIB=$PWD/build/libzotk_phases.so
for v in -1 1 0; do
  ZOT_TUNE=dedupe_variant=$v timeout -k 10 200 python tools/p0_phases.py 50e6 1 25 > gpurun_out/ph_dd_v$v.json 2> gpurun_out/ph_dd_v$v.err || { tail -5 gpurun_out/ph_dd_v$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ph_dd_v$v.json"))
print("variant $v", d.get("dedupe_cycles_per_block"), d.get("rle_ms"))
PY
done
