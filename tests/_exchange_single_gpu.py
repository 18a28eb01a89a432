"""Run by tests/test_gpu_exchange.py in its own process (torch is imported first, as bench.py does for
N > 1): the GPU half of the multi-GPU exchange -- cut points with zk_lower_bound, the k-way merge of
the received pieces with zk_merge_n on torch-owned memory -- emulated for `world` ranks on ONE device,
with plain tensor copies standing in for the all-to-all.  Compared with one oracle run over all reads."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zkoracle as zo            # noqa: E402
from zotmer_amd import native, parallel, synth   # noqa: E402

K, world, R = 25, 4, 3000
kw = dict(genome=40000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.0005))
ctx = native.Context(0)
ops = parallel.GpuOps(ctx)
cuts = parallel.splitters(K, world)

tables = []
for r in range(world):
    d = ctx.synth_reads(synth.DEFAULT_SEED, r * R, R, 150, **kw)
    kt = torch.empty(2 * d.n, dtype=torch.int64, device="cuda")
    ct = torch.empty(2 * d.n, dtype=torch.int32, device="cuda")
    k, c, st = ctx.kmerize(d, K, out=(native.DeviceArray.borrow(ctx, kt.data_ptr(), np.uint64, kt.numel(), keep=kt),
                                      native.DeviceArray.borrow(ctx, ct.data_ptr(), np.uint32, ct.numel(), keep=ct)))
    ctx.sync()
    pos = [0] + ops.lower_bound(kt, k.n, cuts) + [k.n]
    tables.append((kt, ct, pos))

got_k, got_c = [], []
for dst in range(world):
    pieces = [(t[0][t[2][dst]:t[2][dst + 1]], t[1][t[2][dst]:t[2][dst + 1]]) for t in tables]     # what the all-to-all delivers
    rk = torch.cat([p[0] for p in pieces]) if pieces else torch.empty(0, dtype=torch.int64, device="cuda")
    rc = torch.cat([p[1] for p in pieces])
    torch.cuda.synchronize()
    segs, off = [], 0
    for p in pieces:
        segs.append((off, p[0].numel()))
        off += p[0].numel()
    mk, mc = ops.merge_segments(rk, rc, segs)
    got_k.append(mk.to_host())
    got_c.append(mc.to_host())

reads = []
for r in range(world):
    reads += synth.read_strings(synth.DEFAULT_SEED, r * R, R, 150, **kw)
want = zo.kmerize(K, reads)
assert np.array_equal(np.concatenate(got_k), want["kmers"]), "k-mers differ"
assert np.array_equal(np.concatenate(got_c), want["counts"]), "counts differ"
print("EXCHANGE-OK", len(want["kmers"]))
