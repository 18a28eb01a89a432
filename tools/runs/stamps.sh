set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ZOTK_LIB=build/libzotk_stamps.so timeout -k 10 300 python tools/stamps.py 1073741824 3 > gpurun_out/stamps.json 2> gpurun_out/stamps.err || { tail -5 gpurun_out/stamps.err; exit 1; }
cat gpurun_out/stamps.json
