#!/usr/bin/env python3
"""Set-algebra kernels at BASELINE config 3 / config 4 sizes on one MI355X (run on the GPU box).

config 3: zot dist on two sorted sets of 100 M 50-bit k-mers, ~50 % shared   -> zk_project_dedupe + zk_split
config 4 (one GPU's share): merge of 8 x 50 M-k-mer sets drawn from a 200 M pool -> zk_merge_n
plus zk_trim and zk_union_sum on the config 3 sets.  Prints one JSON object; GB/s = algorithmic bytes / time.
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zotmer_amd import native

def timed(ctx, f, reps=5):
    f(); ctx.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ctx.sync(); ts.append(time.perf_counter() - t0)
    return min(ts), r

def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    ctx = native.Context(0)
    rng = np.random.default_rng(3)
    N = int(100_000_000 * scale)
    pool = np.unique(rng.integers(0, 1 << 50, size=int(N * 1.55), dtype=np.uint64))
    sel = rng.random(len(pool))
    A = pool[sel < 0.66][:N]; B = pool[sel > 0.34][:N]
    dA, dB = ctx.upload(A), ctx.upload(B)
    cA = ctx.upload(rng.integers(1, 60, size=len(A), dtype=np.uint32)); cB = ctx.upload(rng.integers(1, 60, size=len(B), dtype=np.uint32))
    out = {"nA": len(A), "nB": len(B)}
    t, abc = timed(ctx, lambda: ctx.split(dA, dB))
    assert abc[0] == len(np.intersect1d(A, B, assume_unique=True))
    out["split"] = dict(ms=t * 1e3, GBps=8 * (len(A) + len(B)) / t / 1e9, abc=abc)
    for sh in (0, 10):
        t, r = timed(ctx, lambda: ctx.project_dedupe(dA, sh))
        out["project_dedupe_shift%d" % sh] = dict(ms=t * 1e3, GBps=(8 * len(A) + 8 * r.n) / t / 1e9, kept=r.n)
    t, r = timed(ctx, lambda: ctx.trim(dA, cA, 3, 40))
    out["trim"] = dict(ms=t * 1e3, GBps=(12 * len(A) + 12 * r[0].n) / t / 1e9, kept=r[0].n)
    t, r = timed(ctx, lambda: ctx.union_sum(dA, cA, dB, cB))
    out["union_sum_u32"] = dict(ms=t * 1e3, GBps=(12 * (len(A) + len(B)) + 12 * r[0].n) / t / 1e9, out=r[0].n)
    t, h = timed(ctx, lambda: ctx.hist(cA))
    out["count_hist"] = dict(ms=t * 1e3, GBps=4 * len(A) / t / 1e9)
    del dA, dB, cA, cB
    # config 4 share: 8 sets of 50 M from a 200 M pool, geometric counts (mean 8)
    M = int(50_000_000 * scale)
    pool = np.unique(rng.integers(0, 1 << 50, size=int(4 * M * 1.02), dtype=np.uint64))[:4 * M]
    sets = []
    for s in range(8):
        k = np.sort(rng.choice(pool, size=M, replace=False)) if scale < 0.2 else pool[rng.random(len(pool)) < 0.25]
        c = rng.geometric(1 / 8.0, size=len(k)).astype(np.uint64)
        sets.append((ctx.upload(k), ctx.upload(c)))
    tot = sum(s[0].n for s in sets)
    mk, mc = ctx.empty(tot, np.uint64), ctx.empty(tot, np.uint64)
    t, r = timed(ctx, lambda: ctx.merge_n(sets, out=(mk, mc)), reps=3)
    out["merge_8x50M"] = dict(ms=t * 1e3, total_in=tot, out=r[0].n, Gpairs_per_s=tot / t / 1e9,
                              GBps_model_3_levels=(3 * 16 * tot + 16 * r[0].n) / t / 1e9)
    print(json.dumps(out, indent=1))

main()
