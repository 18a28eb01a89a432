#!/usr/bin/env python3
"""Micro-benchmark of the radix-sort geometries (run on the GPU box): GB/s of one pass."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zotmer_amd import native

def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 30
    variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
    ctx = native.Context(0)
    groups = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [None]
    # 50-bit pseudo-random keys generated on device: reuse the synth generator bytes as entropy
    rng = np.random.default_rng(1)
    chunk = rng.integers(0, 1 << 50, size=1 << 24, dtype=np.uint64)
    src = ctx.empty(n, np.uint64)
    d = ctx.upload(chunk)
    for off in range(0, n, 1 << 24):
        m = min(1 << 24, n - off)
        ctx._check(ctx.lib.zk_copy(ctx.h, src.ptr + 8 * off, d.ptr, 8 * m))
    ctx.sync()
    # decorrelate the chunks: add offset-dependent constant via a sort-free trick is unnecessary for timing
    work = ctx.empty(n, np.uint64)
    for v, g in [(v, g) for v in variants for g in groups]:
        ctx.tune(sort_variant=v)
        if g is not None:
            ctx.tune(xcd_group=g)
        res = {}
        best = None
        for rep in range(4):
            ctx._check(ctx.lib.zk_copy(ctx.h, work.ptr, src.ptr, 8 * n))
            ctx.sync()
            ctx.profile(True)
            t0 = time.perf_counter()
            ctx.sort_keys(work, 50)
            dt = time.perf_counter() - t0
            res = ctx.profile_read()
            ctx.profile(False)
            pm = res["pass_keys"]["ms"] / res["pass_keys"]["launches"]
            if best is None or pm < best[0]:
                best = (pm, res, dt)
        pm, res, dt = best
        if True:
            pass
        pk = res["pass_keys"]
        h = work.to_host(1 << 20)
        ok = bool(np.all(h[1:] >= h[:-1]))
        print(json.dumps(dict(variant=v, xcd_group=g, n=n, passes=pk["launches"], pass_ms=pk["ms"] / pk["launches"],
                              pass_GBps=pk["bytes"] / 1e9 / (pk["ms"] / 1e3), hist_ms=res["hist_array"]["ms"],
                              total_ms=dt * 1e3, sorted_prefix=ok)), flush=True)

main()
