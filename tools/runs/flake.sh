cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2 3 4; do
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multigpu.py -x -q -k "merge and not merge_n_random" > gpurun_out/fl_a$i.log 2>&1; echo "without kway tests run $i rc=$?"
done
for i in 1 2 3 4; do
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multigpu.py -x -q -k "merge" > gpurun_out/fl_b$i.log 2>&1; echo "with kway tests run $i rc=$?"
done
