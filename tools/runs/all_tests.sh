set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; rc=$?
tail -15 gpurun_out/t_all.log
exit $rc
