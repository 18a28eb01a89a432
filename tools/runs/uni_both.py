import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zotmer_amd import native, synth
ctx = native.Context(0)
cfg = synth.CONFIGS["config2"]
R, L, K = 20_000_000, 150, 25
uni = ctx.synth_reads(synth.DEFAULT_SEED + 1, 0, R, L, genome=0, sub_thr=0, n_thr=synth.frac32(cfg["n"]))
cap = 2 * R * (L - K + 1) + 1024
out = (ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32))
res = {}
for name, flags in (("canonical+mirror", 0), ("both strands sorted", native.KMERIZE_BOTH)):
    k, c, st = ctx.kmerize(uni, K, flags, out=out); ctx.sync()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(2):
        k, c, st = ctx.kmerize(uni, K, flags, out=out)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 2
    prof = ctx.profile_read(); ctx.profile(False)
    res[name] = dict(ms=dt * 1e3, unique=st.n_unique, ok=bool(ctx.checksum(k, c) == ctx.stream_checksum(uni, K)),
                     kernels={n: round(v["ms"] / 2, 2) for n, v in prof.items()})
print(json.dumps(res, indent=1))
