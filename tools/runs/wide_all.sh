set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in 1; do
ZK_WIDE_ALL=$v ZOTK_LIB=build/libzotk_phases.so timeout -k 10 300 python - > gpurun_out/wide_all_$v.json 2> gpurun_out/wide_all_$v.err <<PY || { tail -5 gpurun_out/wide_all_$v.err; exit 1; }
import json, numpy as np
from zotmer_amd import native, synth
cfg = synth.CONFIGS["config2"]
ctx = native.Context(0)
R = cfg["reads"]
d = ctx.synth_reads(synth.DEFAULT_SEED, 0, R, cfg["L"], genome=cfg["genome"], sub_thr=synth.frac32(cfg["sub"]), n_thr=synth.frac32(cfg["n"]))
cap = int(2 * (cfg["genome"] + R * cfg["L"] * cfg["sub"] * 22) * 1.25) + (1 << 20)
outs = (ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32))
res = []
for rep in range(3):
    ctx.profile(True)
    ctx.kmerize(d, 25, out=outs)
    ctx.sync()
    p = ctx.profile_read(); ctx.profile(False)
    res.append({k: round(v["ms"], 2) for k, v in p.items() if v["launches"]})
print(json.dumps({"local": $v, "runs": res}))
PY
cat gpurun_out/wide_all_$v.json
done
