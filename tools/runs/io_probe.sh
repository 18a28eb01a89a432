cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
df -h . /tmp /dev/shm 2>/dev/null | head -5
mount | grep -E " / | /tmp |overlay" | head -3
for d in "$GRAFT_REPO_ROOT/gpurun_out" /tmp /dev/shm; do
  echo "== $d"
  ( time dd if=/dev/zero of=$d/io_probe.bin bs=16M count=256 conv=fsync 2>&1 | tail -1 ) 2>&1 | grep -E "copied|real"
  ( time dd if=/dev/zero of=$d/io_probe.bin bs=16M count=256 oflag=direct 2>&1 | tail -1 ) 2>&1 | grep -E "copied|real|Invalid"
  rm -f $d/io_probe.bin
done
nproc; free -g | head -2
