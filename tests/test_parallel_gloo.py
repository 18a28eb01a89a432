"""
The multi-GPU protocol (zotmer_amd/parallel.py) under the gloo backend on CPU tensors, world_size 2 and 3: balanced
value-range splitters and the hash-range owner, the all-to-all rounds, the merge of the received pieces, and the
reductions -- through the PRODUCT functions Exchange.exchange_and_merge / merge_sets / dist_pair / gather_to_root that
`zot kmerize | merge | dist` and bench.py call.  The data-path arithmetic is the CPU oracle here (on the GPU it is
libzotk); results are compared with one oracle run over all inputs.
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
    """Same interface as parallel.GpuOps, on CPU tensors, arithmetic by the oracle."""

    def empty(self, n, dtype):
        return torch.empty(max(int(n), 1), dtype=dtype)

    def lower_bound(self, keys_t, n, cuts):
        k = keys_t[:n].numpy().view(np.uint64)
        return np.searchsorted(k, np.asarray(cuts, dtype=np.uint64), side="left").astype(np.uint64)

    def hash_partition(self, keys_t, counts_t, n, world, seed):
        k = keys_t[:n].numpy().view(np.uint64)
        own = np.array([parallel.hash_owner(x, seed, world, zo.murmer) for x in k], dtype=np.int64)
        order = np.argsort(own, kind="stable")
        offs = [0] + [int(v) for v in np.cumsum(np.bincount(own, minlength=world))]
        ok = torch.from_numpy(k[order].view(np.int64).copy()) if n else self.empty(0, torch.int64)
        oc = None
        if counts_t is not None:
            oc = counts_t[:n][torch.from_numpy(order)].clone() if n else self.empty(0, counts_t.dtype)
        return ok, oc, offs

    def before_comm(self):
        pass

    def after_comm(self):
        pass

    def merge_segments(self, keys_t, counts_t, segs, want_acgt=False):
        k = np.empty(0, np.uint64)
        c = np.empty(0, np.uint64)
        for o, n in segs:
            sk = keys_t[o:o + n].numpy().view(np.uint64)
            raw = counts_t[o:o + n].numpy()
            sc = raw.view(np.uint32).astype(np.uint64) if raw.dtype == np.int32 else raw.view(np.uint64)
            assert np.all(sk[1:] > sk[:-1])
            k, c = zo.union_sum(k, c, sk, sc)
        if want_acgt:
            acgt = [int(c[(k & np.uint64(3)) == b].sum()) for b in range(4)]
            return k, c, acgt
        return k, c

    def dedupe(self, keys_t, n, shift=0):
        k = keys_t[:n].numpy().view(np.uint64) >> np.uint64(shift)
        keep = np.ones(len(k), dtype=bool)
        keep[1:] = k[1:] != k[:-1]
        out = k[keep]
        return torch.from_numpy(out.view(np.int64).copy()) if len(out) else self.empty(0, torch.int64), len(out)

    def split(self, x_t, nx, y_t, ny):
        return zo.split(x_t[:nx].numpy().view(np.uint64), y_t[:ny].numpy().view(np.uint64))

    def checksum(self, k, c):
        m = (1 << 64) - 1
        s0 = int(c.sum()) & m
        s1 = sum(int(a) * int(b) for a, b in zip(k, c)) & m
        s2 = sum(zo.murmer(int(a), 0) * int(b) for a, b in zip(k, c)) & m
        return (s0, s1, s2)

    def hist(self, c):
        v, f = np.unique(c, return_counts=True)
        return {int(a): int(b) for a, b in zip(v, f)}

    def to_tensors(self, k, c):
        return torch.from_numpy(k.view(np.int64).copy()), torch.from_numpy(c.view(np.int64).copy()), len(k)

    def mirror_expand(self, k, c, K):
        """both strands of a counted canonical list: (x, n) and (rc x, n), a palindrome counted n + n"""
        acc = {}
        for x, n in zip(k, c):
            x, n = int(x), int(n)
            acc[x] = acc.get(x, 0) + n
            y = zo.rc(K, x)
            acc[y] = acc.get(y, 0) + n
        ks = np.array(sorted(acc), dtype=np.uint64)
        return ks, np.array([acc[int(x)] for x in ks], dtype=np.uint64)


def canonical_of(table, K):
    """the counted canonical list zk_kmerize(ZK_KMERIZE_CANONICAL_ONLY) returns, from the oracle's both-strand table"""
    k, c = table["kmers"], table["counts"]
    rc = np.array([zo.rc(K, int(x)) for x in k], dtype=np.uint64)
    keep = k <= rc
    cc = c[keep].astype(np.uint64)
    cc[k[keep] == rc[keep]] //= 2                      # a palindrome was emitted twice per window
    return k[keep], cc.astype(np.uint32)


def _reads(rank):
    return synth.read_strings(synth.DEFAULT_SEED, rank * READS_PER_RANK, READS_PER_RANK, 150, **KW)


def _skewed_reads(rank):
    """mostly AT-rich and low-complexity reads (an AT-rich genome, poly-A tails, AC repeats): their k-mers start with A or
    T, so equal-width value ranges starve the middle ranks"""
    rs = _reads(rank)[:100]
    rng = np.random.default_rng(100 + rank)
    for i in range(500):
        if i % 5 == 0:
            s = np.array(list("AC" * 75))
        elif i % 5 == 1:
            s = np.array(list("A" * 150))
        else:
            s = np.array(list("AT"))[rng.integers(0, 2, size=150)]
        for p in rng.integers(0, 150, size=3):
            s[p] = "ACGT"[rng.integers(0, 4)]
        rs.append("".join(s))
    return rs


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _kmerize_worker(rank, world, port, outdir, owner, chunk, skew, canonical=False):
    _init(rank, world, port)
    try:
        reads = _skewed_reads(rank) if skew else _reads(rank)
        local = zo.kmerize(K, reads)
        ex = parallel.Exchange(None, dist, K, ops=NumpyOps(), owner=owner, seed=7)
        if chunk:
            ex.CHUNK = chunk          # force several rounds per all-to-all (on the GPU a round is <= 256 MiB per peer)
        if canonical:
            # the product flow of `zot kmerize` / bench.py: canonical lists travel, the owner rebuilds the strands
            ck, cc = canonical_of(local, K)
            kt = torch.from_numpy(ck.view(np.int64).copy())
            ct = torch.from_numpy(cc.view(np.int32).copy())
            k, c = ex.kmerize_finish(kt, ct, len(ck))
            assert all((min(int(x), zo.rc(K, int(x))) >= (([0] + ex.cuts + [1 << (2 * K)])[rank]) if owner == "range" else True) for x in k[:100])
        else:
            n = len(local["kmers"])
            kt = torch.from_numpy(local["kmers"].view(np.int64).copy())
            ct = torch.from_numpy(local["counts"].view(np.int32).copy())
            if owner == "range":
                cuts = ex.balanced_cuts([(kt, n)])
            k, c = ex.exchange_and_merge(kt, ct, n)
        if canonical:
            pass
        elif owner == "range":          # every k-mer this rank now owns lies in its value range
            edges = [0] + cuts + [1 << (2 * K)]
            assert len(k) == 0 or (int(k[0]) >= edges[rank] and int(k[-1]) < edges[rank + 1])
        else:
            assert all(parallel.hash_owner(x, 7, world, zo.murmer) == rank for x in k[:200])
        # order-free check across ranks, as bench.py does
        inst = np.concatenate([zo.kmers_list(K, r, True) for r in reads])
        m = (1 << 64) - 1
        stream_sums = (len(inst) & m, sum(int(x) for x in inst) & m, sum(zo.murmer(int(x), 0) for x in inst) & m)
        assert ex.verify_global(k, c, stream_sums)
        gk, gc = ex.gather_to_root(k, c)
        if rank == 0:
            np.savez(os.path.join(outdir, "root.npz"), k=gk, c=gc)
        np.savez(os.path.join(outdir, "r%d.npz" % rank), k=k, c=c)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,owner,chunk", [(2, "range", None), (3, "range", 1000), (2, "hash", 7001), (3, "hash", None)])
def test_kmerize_exchange_gloo(tmp_path, world, owner, chunk):
    mp.spawn(_kmerize_worker, args=(world, _free_port(), str(tmp_path), owner, chunk, False), nprocs=world, join=True)
    parts = [np.load(str(tmp_path / ("r%d.npz" % r))) for r in range(world)]
    want = zo.kmerize(K, [s for r in range(world) for s in _reads(r)])
    root = np.load(str(tmp_path / "root.npz"))
    assert np.array_equal(root["k"], want["kmers"])                   # what the writer of the output file holds
    assert np.array_equal(root["c"], want["counts"].astype(np.uint64))
    if owner == "range":                                               # concatenation in rank order IS the global sorted set
        assert np.array_equal(np.concatenate([p["k"] for p in parts]), want["kmers"])
        assert np.array_equal(np.concatenate([p["c"] for p in parts]), want["counts"].astype(np.uint64))
    sizes = [len(p["k"]) for p in parts]
    assert max(sizes) <= 1.2 * (sum(sizes) / world)


@pytest.mark.parametrize("world,owner,chunk", [(2, "range", 900), (3, "hash", None), (3, "range", None)])
def test_kmerize_canonical_exchange_gloo(tmp_path, world, owner, chunk):
    """Exchange.kmerize_finish: the counted CANONICAL lists are exchanged and every rank rebuilds both strands of what it owns"""
    mp.spawn(_kmerize_worker, args=(world, _free_port(), str(tmp_path), owner, chunk, False, True), nprocs=world, join=True)
    parts = [np.load(str(tmp_path / ("r%d.npz" % r))) for r in range(world)]
    want = zo.kmerize(K, [s for r in range(world) for s in _reads(r)])
    root = np.load(str(tmp_path / "root.npz"))
    assert np.array_equal(root["k"], want["kmers"])
    assert np.array_equal(root["c"], want["counts"].astype(np.uint64))
    assert sum(len(p["k"]) for p in parts) == len(want["kmers"])          # the pieces partition the table
    sizes = [len(p["k"]) for p in parts]
    assert max(sizes) <= 1.2 * (sum(sizes) / world)


@pytest.mark.parametrize("owner", ["range", "hash"])
def test_balanced_on_skewed_reads_gloo(tmp_path, owner):
    """Half the reads poly-A / AC repeats: both owners must keep max / mean owned size <= 1.2 and give the oracle's table
    (equal-width ranges put > 60 % of this input on rank 0)."""
    world = 3
    mp.spawn(_kmerize_worker, args=(world, _free_port(), str(tmp_path), owner, None, True), nprocs=world, join=True)
    parts = [np.load(str(tmp_path / ("r%d.npz" % r))) for r in range(world)]
    want = zo.kmerize(K, [s for r in range(world) for s in _skewed_reads(r)])
    root = np.load(str(tmp_path / "root.npz"))
    assert np.array_equal(root["k"], want["kmers"])
    assert np.array_equal(root["c"], want["counts"].astype(np.uint64))
    sizes = [len(p["k"]) for p in parts]
    assert max(sizes) <= 1.2 * (sum(sizes) / world), sizes
    static = parallel.splitters(K, world)
    pos = [0] + list(np.searchsorted(want["kmers"], np.array(static, dtype=np.uint64))) + [len(want["kmers"])]
    assert max(np.diff(pos)) > 1.2 * len(want["kmers"]) / world        # the static cuts would NOT have balanced it


def _merge_dist_worker(rank, world, port, outdir, owner):
    """`zot merge` of 6 sets and `zot dist` of two sets over `world` ranks through the product functions."""
    _init(rank, world, port)
    try:
        ops = NumpyOps()
        ex = parallel.Exchange(None, dist, K, ops=ops, owner=owner, seed=3)
        # merge: rank r loads sets r, r + world, ... and merges them locally (on the GPU: zk_merge_n)
        k = np.empty(0, np.uint64)
        c = np.empty(0, np.uint64)
        for s in range(rank, 6, world):
            t = zo.kmerize(K, _reads(s))
            k, c = zo.union_sum(k, c, t["kmers"], t["counts"].astype(np.uint64))
        kt = torch.from_numpy(k.view(np.int64).copy())
        ct = torch.from_numpy(c.view(np.int64).copy())
        res = ex.merge_sets(kt, ct, len(k))
        gk, gc = ex.gather_to_root(res["k"], res["c"])
        # dist, K = 25 and projected to K = 11: every rank holds the r-th contiguous chunk of each sorted set
        a = zo.kmerize(K, _reads(0))["kmers"]
        b = zo.kmerize(K, _reads(1) + _reads(0)[:300])["kmers"]
        out = {}
        for kk in (25, 11):
            exd = parallel.Exchange(None, dist, kk, ops=ops, owner=owner, seed=3)
            ca, cb = np.array_split(a, world)[rank], np.array_split(b, world)[rank]
            at = torch.from_numpy(ca.view(np.int64).copy())
            bt = torch.from_numpy(cb.view(np.int64).copy())
            abc, sizes = exd.dist_pair(at, len(ca), bt, len(cb), shift=2 * (K - kk))
            out["abc%d" % kk] = np.array(abc)
            out["sz%d" % kk] = np.array(sizes)
        np.savez(os.path.join(outdir, "m%d.npz" % rank), acgt=np.array(res["acgt"], dtype=np.uint64),
                 hv=np.array(sorted(res["hist"]), dtype=np.uint64), hf=np.array([res["hist"][v] for v in sorted(res["hist"])], dtype=np.uint64),
                 ng=res["n_global"], **out)
        if rank == 0:
            np.savez(os.path.join(outdir, "mroot.npz"), k=gk, c=gc)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,owner", [(2, "range"), (3, "range"), (3, "hash")])
def test_distributed_merge_and_dist_gloo(tmp_path, world, owner):
    mp.spawn(_merge_dist_worker, args=(world, _free_port(), str(tmp_path), owner), nprocs=world, join=True)
    parts = [np.load(str(tmp_path / ("m%d.npz" % r))) for r in range(world)]
    root = np.load(str(tmp_path / "mroot.npz"))
    zs, zc, zacgt = zo.merge_n(K, [(t["kmers"], t["counts"].astype(np.uint64)) for t in (zo.kmerize(K, _reads(s)) for s in range(6))])
    assert np.array_equal(root["k"], zs)
    assert np.array_equal(root["c"], zc)
    hv, hf = zo.hist(zc)
    a = zo.kmerize(K, _reads(0))["kmers"]
    b = zo.kmerize(K, _reads(1) + _reads(0)[:300])["kmers"]
    for p in parts:
        assert [int(v) for v in p["acgt"]] == [int(v) for v in zacgt]
        assert np.array_equal(p["hv"], hv) and np.array_equal(p["hf"], hf)
        assert int(p["ng"]) == len(zs)
        for kk in (25, 11):
            pa, pb = zo.project_dedupe(a, 2 * (K - kk)), zo.project_dedupe(b, 2 * (K - kk))
            assert tuple(int(v) for v in p["abc%d" % kk]) == zo.split(pa, pb)
            assert tuple(int(v) for v in p["sz%d" % kk]) == (len(pa), len(pb))


def test_all_reduce_u64_gloo(tmp_path):
    mp.spawn(_reduce_worker, args=(3, _free_port()), nprocs=3, join=True)


def _reduce_worker(rank, world, port):
    _init(rank, world, port)
    try:
        comm = parallel.TorchComm(dist)
        big = (1 << 64) - 5
        assert comm.all_reduce([big, rank, 1 << 40]) == [(3 * big) & parallel.M64, 3, 3 << 40]
        assert comm.all_reduce([big - rank, rank + (7 << 33)], "max") == [big, 2 + (7 << 33)]
        arr = comm.all_reduce(np.array([rank + 1, 1 << 63], dtype=np.uint64))
        assert arr.dtype == np.uint64 and int(arr[0]) == 6 and int(arr[1]) == (3 << 63) & parallel.M64
    finally:
        dist.destroy_process_group()


def test_splitters():
    assert parallel.splitters(25, 1) == []
    assert parallel.splitters(2, 4) == [4, 8, 12]
    s = parallel.splitters(31, 8)
    assert len(s) == 7 and all(b > a for a, b in zip(s, s[1:])) and s[-1] < (1 << 62)
