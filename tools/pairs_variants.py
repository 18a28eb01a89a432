#!/usr/bin/env python3
"""bench.py-like timing of zk_kmerize on N reads for several (sort_variant, pairs_variant) choices."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zotmer_amd import native, synth
R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
ctx = native.Context(0)
cfg = synth.CONFIGS["config2"]
d = ctx.synth_reads(synth.DEFAULT_SEED, 0, R, 150, genome=cfg["genome"], sub_thr=synth.frac32(cfg["sub"]), n_thr=synth.frac32(cfg["n"]))
cap = int(R * 151 * 0.35) + (1 << 20)
ok, oc = ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32)
for sv, pv, ss in ((3, 2, 0),):
    ctx.tune(sort_variant=sv, pairs_variant=pv, short_sort=ss)
    ctx.kmerize(d, 25, out=(ok, oc))
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(2):
        k, c, st = ctx.kmerize(d, 25, out=(ok, oc))
    ctx.sync()
    dt = (time.perf_counter() - t0) / 2
    pr = ctx.profile_read(); ctx.profile(False)
    print(json.dumps(dict(sort=sv, pairs=pv, short=ss, ms=dt * 1e3, Gkps=st.n_instances / dt / 1e9,
                          kernels={k: round(v["ms"] / 2, 1) for k, v in pr.items()})), flush=True)
