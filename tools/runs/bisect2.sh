for rev in old_r03 t_ebd3ad4; do
  cd "$GRAFT_REPO_ROOT/build/$rev" || exit 1
  f=0
  for i in $(seq 1 30); do
    timeout -k 10 300 python tests/_logical_ranks_gpu.py > ../../gpurun_out/bis2_${rev}_$i.log 2>&1 || f=$((f+1))
  done
  echo "$rev: $f of 30 failed (standalone script)"
done
