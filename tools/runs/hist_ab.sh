set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export ZOTK_LIB=$PWD/build/libzotk_phases.so
for n in 2 1 0; do
  ZK_HIST_ATOMICS=$n timeout -k 10 200 python tools/p0_phases.py 50e6 1 25 > gpurun_out/ph_hist_$n.json 2> gpurun_out/ph_hist_$n.err || { tail -5 gpurun_out/ph_hist_$n.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ph_hist_$n.json"))
print("atomics $n: hist_stream_ms", d.get("hist_stream_ms"))
PY
done
