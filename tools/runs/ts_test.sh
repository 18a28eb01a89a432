set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sort" > gpurun_out/ts_test.log 2>&1; rc=$?
tail -15 gpurun_out/ts_test.log
exit $rc
