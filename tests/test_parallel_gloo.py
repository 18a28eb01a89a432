"""
The multi-GPU exchange protocol (zotmer_amd/parallel.py) under the gloo backend on CPU tensors,
world_size 2 and 3: value-range cut points, the two all-to-all rounds, the merge of the received
pieces, and the checksum / (a,b,c) reductions.  The data-path arithmetic is the CPU oracle here (on
the GPU it is libzotk); the result is compared with one oracle run over all reads.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import zkoracle as zo
from zotmer_amd import parallel, synth

K = 25
READS_PER_RANK = 1200
KW = dict(genome=20000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.0005))


class NumpyOps:
    """Same interface as parallel.GpuOps, on CPU tensors."""

    def empty(self, n, dtype):
        return torch.empty(max(int(n), 1), dtype=dtype)

    def lower_bound(self, keys_t, n, cuts):
        k = keys_t[:n].numpy().view(np.uint64)
        return [int(p) for p in np.searchsorted(k, np.array(cuts, dtype=np.uint64), side="left")]

    def before_comm(self):
        pass

    def after_comm(self):
        pass

    def merge_segments(self, keys_t, counts_t, segs):
        k = np.empty(0, np.uint64)
        c = np.empty(0, np.uint64)
        for o, n in segs:
            sk = keys_t[o:o + n].numpy().view(np.uint64)
            sc = counts_t[o:o + n].numpy().view(np.uint32).astype(np.uint64)
            assert np.all(sk[1:] > sk[:-1])
            k, c = zo.union_sum(k, c, sk, sc)
        return k, c

    def checksum(self, k, c):
        m = (1 << 64) - 1
        s0 = int(c.sum()) & m
        s1 = sum(int(a) * int(b) for a, b in zip(k, c)) & m
        s2 = sum(zo.murmer(int(a), 0) * int(b) for a, b in zip(k, c)) & m
        return (s0, s1, s2)


def _reads(rank):
    return synth.read_strings(synth.DEFAULT_SEED, rank * READS_PER_RANK, READS_PER_RANK, 150, **KW)


def _worker(rank, world, port, outdir, chunk=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = zo.kmerize(K, _reads(rank))
        n = len(local["kmers"])
        kt = torch.from_numpy(local["kmers"].view(np.int64).copy())
        ct = torch.from_numpy(local["counts"].view(np.int32).copy())
        ex = parallel.RangeExchange(None, dist, K, ops=NumpyOps())
        if chunk:
            ex.CHUNK = chunk          # force several rounds per all-to-all (on the GPU a round is <= 256 MiB per peer)
        k, c = ex.exchange_and_merge(kt, ct, n)
        # every k-mer this rank now owns lies in its value range
        cuts = [0] + parallel.splitters(K, world) + [1 << (2 * K)]
        assert len(k) == 0 or (int(k[0]) >= cuts[rank] and int(k[-1]) < cuts[rank + 1])
        # order-free check across ranks, as bench.py --verify does
        inst = np.concatenate([zo.kmers_list(K, r, True) for r in _reads(rank)])
        m = (1 << 64) - 1
        stream_sums = (len(inst) & m, sum(int(x) for x in inst) & m, sum(zo.murmer(int(x), 0) for x in inst) & m)
        assert ex.verify_global(k, c, stream_sums)
        # dist: (a, b, c) of the rank's range against a shifted copy, all-reduced
        other = k[::2]
        abc = ex.split_counts(zo.split(k, other))
        np.savez(os.path.join(outdir, "r%d.npz" % rank), k=k, c=c, abc=np.array(abc))
    finally:
        dist.destroy_process_group()


def _merge_dist_worker(rank, world, port, outdir):
    """`zot merge` and `zot dist` over `world` ranks: rank r owns sets r, r+world, ... (64-bit counts)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ops = NumpyOps()
        ex = parallel.RangeExchange(None, dist, K, ops=ops)
        mine = [zo.kmerize(K, _reads(s)) for s in range(rank, 6, world)]
        k = np.empty(0, np.uint64); c = np.empty(0, np.uint64)
        for t in mine:                                     # local k-way merge of this rank's sets
            k, c = zo.union_sum(k, c, t["kmers"], t["counts"].astype(np.uint64))
        kt = torch.from_numpy(k.view(np.int64).copy()); ct = torch.from_numpy(c.view(np.int64).copy())
        rk, rc, segs = ex.exchange(kt, ct, len(k))
        mk = np.empty(0, np.uint64); mc = np.empty(0, np.uint64)
        for o, n in segs:
            mk, mc = zo.union_sum(mk, mc, rk[o:o + n].numpy().view(np.uint64), rc[o:o + n].numpy().view(np.uint64))
        # dist: every rank holds both sets in full and splits only its value range
        a = zo.kmerize(K, _reads(0))["kmers"]; b = zo.kmerize(K, _reads(1))["kmers"]
        at = torch.from_numpy(a.view(np.int64).copy()); bt = torch.from_numpy(b.view(np.int64).copy())
        a0, a1 = ex.owned_slice(at, len(a)); b0, b1 = ex.owned_slice(bt, len(b))
        abc = ex.split_counts(zo.split(a[a0:a1], b[b0:b1]))
        np.savez(os.path.join(outdir, "m%d.npz" % rank), k=mk, c=mc, abc=np.array(abc))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_merge_and_dist_gloo(tmp_path, world):
    mp.spawn(_merge_dist_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(str(tmp_path / ("m%d.npz" % r))) for r in range(world)]
    zs, zc, _ = zo.merge_n(K, [(t["kmers"], t["counts"].astype(np.uint64)) for t in (zo.kmerize(K, _reads(s)) for s in range(6))])
    assert np.array_equal(np.concatenate([p["k"] for p in parts]), zs)
    assert np.array_equal(np.concatenate([p["c"] for p in parts]), zc)
    want = zo.split(zo.kmerize(K, _reads(0))["kmers"], zo.kmerize(K, _reads(1))["kmers"])
    for p in parts:
        assert tuple(p["abc"]) == want


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,chunk", [(2, None), (3, None), (2, 7001), (3, 1000)])
def test_range_exchange_gloo(tmp_path, world, chunk):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), chunk), nprocs=world, join=True)
    parts = [np.load(str(tmp_path / ("r%d.npz" % r))) for r in range(world)]
    k = np.concatenate([p["k"] for p in parts])
    c = np.concatenate([p["c"] for p in parts])
    want = zo.kmerize(K, [s for r in range(world) for s in _reads(r)])
    assert np.array_equal(k, want["kmers"])                       # concatenation in rank order IS the global sorted set
    assert np.array_equal(c, want["counts"].astype(np.uint64))
    a = sum(len(p["k"][::2]) for p in parts)
    assert tuple(parts[0]["abc"]) == (a, len(k) - a, 0)


def test_splitters():
    assert parallel.splitters(25, 1) == []
    assert parallel.splitters(2, 4) == [4, 8, 12]
    s = parallel.splitters(31, 8)
    assert len(s) == 7 and all(b > a for a, b in zip(s, s[1:])) and s[-1] < (1 << 62)
