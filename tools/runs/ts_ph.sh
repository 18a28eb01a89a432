set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "early_collapse or large_without or canonical_only" > gpurun_out/ts_test2.log 2>&1 || { tail -30 gpurun_out/ts_test2.log; exit 1; }
tail -2 gpurun_out/ts_test2.log
ZOTK_LIB=build/libzotk_phases.so timeout -k 10 300 python tools/ts_phases.py 2e7 25 > gpurun_out/ts_phases.json 2> gpurun_out/ts_phases.err || { tail -5 gpurun_out/ts_phases.err; exit 1; }
cat gpurun_out/ts_phases.json
