for rev in 1ec97be 02d55c3 ebd3ad4; do
  cd "$GRAFT_REPO_ROOT/build/t_$rev" || exit 1
  f=0
  for i in 1 2 3 4 5 6 7 8; do
    timeout -k 10 300 python -m pytest tests/test_gpu_multigpu.py -x -q -k "logical" > ../../gpurun_out/bis_${rev}_$i.log 2>&1 || f=$((f+1))
  done
  echo "$rev: $f of 8 failed"
done
