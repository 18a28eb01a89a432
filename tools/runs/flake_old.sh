cd "$GRAFT_REPO_ROOT/build/old_r03" || exit 1
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multigpu.py -x -q -k "merge" > ../../gpurun_out/fl_old$i.log 2>&1; echo "r03 tree run $i rc=$?"
done
