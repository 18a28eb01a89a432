set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ZOT_TIMING=2 timeout -k 10 400 python bench.py --no-cpu-baseline --only-extra config2_e2e_h2d > gpurun_out/e2e_h2d.json 2> gpurun_out/e2e_h2d.err || { tail -15 gpurun_out/e2e_h2d.err; exit 1; }
cat gpurun_out/e2e_h2d.json
grep engine gpurun_out/e2e_h2d.err | tail -45
