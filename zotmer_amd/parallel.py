"""
Multi-GPU exchange step: one process per GPU, `torch.distributed` with
the "nccl" backend (= RCCL over xGMI on ROCm).

kmerize shards READS: every rank counts its own reads with no communication at all.  The per-rank
tables then meet in ONE exchange: the k-mer value space [0, 4**K) is cut into `world` contiguous
ranges, each rank sends range r of its sorted table to rank r with a single all-to-all-v (a sorted
table is already partitioned: the cut points are binary searches, no scatter), and every rank
union-sums the `world` sorted pieces it receives.  Rank r then owns the r-th contiguous range of the
global table, so the global sorted set is the concatenation of the ranks' tables in rank order.
xGMI is a full mesh: an all-to-all is one hop per peer and per-link bound.

`dist` (a, b, c) counts are additive over any partition of the key space: partition both sets with the
same splitters, zk_split locally, all-reduce three integers (`split_counts`).

The arithmetic on the data path is injected (`ops`), so the protocol -- split sizes, the two
all-to-all rounds, merge order, the checksum reduction -- runs unchanged under the "gloo" backend on
CPU tensors in the tests.

`zot merge` over N GPUs = the same exchange: every rank merges the sets it loaded (zk_merge_n), the
result is range-partitioned and exchanged, and the received pieces are merged again; the global set
is the concatenation of the ranks' pieces in rank order.
"""
import os

import numpy as np
import torch


def splitters(K, world):
    """world-1 ascending cut points of [0, 4**K): range r is [cut[r-1], cut[r])."""
    space = 1 << (2 * K)
    return [(space * r) // world for r in range(1, world)]


class GpuOps:
    """Data-path operations on torch CUDA tensors through libzotk (the tensors only carry bytes:
    uint64 k-mers travel as int64, uint32 counts as int32)."""

    def __init__(self, ctx):
        self.ctx = ctx
        from zotmer_amd import native
        self._borrow = native.DeviceArray.borrow

    def empty(self, n, dtype):
        return torch.empty(max(int(n), 1), dtype=dtype, device="cuda")

    def k_array(self, t, n, off=0):
        return self._borrow(self.ctx, t.data_ptr() + 8 * off, np.uint64, n, keep=t)

    def c_array(self, t, n, off=0):
        if t.dtype == torch.int64:       # uint64 counts (merge) travel as int64, uint32 counts (kmerize) as int32
            return self._borrow(self.ctx, t.data_ptr() + 8 * off, np.uint64, n, keep=t)
        return self._borrow(self.ctx, t.data_ptr() + 4 * off, np.uint32, n, keep=t)

    def lower_bound(self, keys_t, n, cuts):
        return self.ctx.lower_bound(self.k_array(keys_t, n), cuts)

    def before_comm(self):
        self.ctx.sync()                 # our kernels run on the ctx's own stream

    def after_comm(self):
        torch.cuda.synchronize()

    def merge_segments(self, keys_t, counts_t, segs):
        """k-way union-sum of the sorted segments [(offset, length)] of the receive buffers
        (zk_merge_n: a tree of merge-path passes inside the library's workspace; the output buffers
        are kept between steps, so a steady-state step allocates nothing)."""
        parts = [(self.k_array(keys_t, n, o), self.c_array(counts_t, n, o)) for o, n in segs if n]
        if not parts:
            return self.k_array(keys_t, 0), self.c_array(counts_t, 0)
        if len(parts) == 1:
            return parts[0]
        total = sum(p[0].n for p in parts)
        cdt = parts[0][1].dtype
        if getattr(self, "_mk", None) is None or self._mk.n < total or self._mc.dtype != cdt:
            self._mk = self._mc = None
            self._mk = self.ctx.empty(total + total // 16, np.uint64)
            self._mc = self.ctx.empty(total + total // 16, cdt)
        k, c, _ = self.ctx.merge_n(parts, out=(self._mk, self._mc))
        return k, c

    def checksum(self, k, c):
        return self.ctx.checksum(k, c)


class RangeExchange:
    def __init__(self, ctx, dist, K, ops=None):
        self.dist, self.K = dist, K
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.ops = ops if ops is not None else GpuOps(ctx)
        self.cuts = splitters(K, self.world)
        self._rk = self._rc = None        # receive buffers, kept between steps (grown on demand)

    def exchange(self, keys_t, counts_t, n):
        """keys_t (int64) / counts_t (int32): this rank's sorted table in its first n entries.
        Returns (recv keys, recv counts, [(offset, length) per source rank])."""
        ops, dist, W = self.ops, self.dist, self.world
        pos = [0] + ops.lower_bound(keys_t, n, self.cuts) + [n]
        send = [pos[r + 1] - pos[r] for r in range(W)]
        ops.before_comm()
        s = torch.tensor(send, dtype=torch.int64, device=keys_t.device)
        r = torch.empty(W, dtype=torch.int64, device=keys_t.device)
        dist.all_to_all_single(r, s)
        recv = [int(v) for v in r.tolist()]
        total = sum(recv)
        if self._rk is None or self._rk.numel() < total or self._rc.dtype != counts_t.dtype:
            self._rk = self._rc = None
            self._rk = ops.empty(total + total // 16, torch.int64)
            self._rc = ops.empty(total + total // 16, counts_t.dtype)
        rk, rc = self._rk, self._rc
        self._all_to_all_v(rk, keys_t, recv, send, pos)
        self._all_to_all_v(rc, counts_t, recv, send, pos)
        ops.after_comm()
        segs, off = [], 0
        for m in recv:
            segs.append((off, m))
            off += m
        return rk, rc, segs

    # elements per peer and round.  One all_to_all_single with a per-peer message above 1 GiB arrives with its second
    # half wrong on this stack (RCCL 2.26.6 / torch 2.10, measured with a one-rank self exchange: 2^27 int64 fine,
    # 2^27 + 1 corrupt), and config 2 on 8 GPUs sends 1.6 GB per peer -- so every message is cut to <= 256 MiB.
    CHUNK = 1 << 25

    def _all_to_all_v(self, out_t, in_t, recv, send, pos):
        """out_t[roff[r] : roff[r] + recv[r]] <- rank r's in_t[pos[me] : pos[me + 1]], in rounds of at most CHUNK
        elements per peer.  The slices of a round are not contiguous, so they go through two staging buffers."""
        dist, W, CH = self.dist, self.world, self.CHUNK
        roff = [0]
        for m in recv:
            roff.append(roff[-1] + m)
        biggest = max([0] + list(send) + list(recv))
        rounds = (biggest + CH - 1) // CH
        if W > 1:
            t = torch.tensor([rounds], dtype=torch.int64, device=in_t.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rounds = int(t.item())
        if rounds <= 1:
            dist.all_to_all_single(out_t[:roff[-1]], in_t[:pos[-1]], recv, send)
            return
        key = (in_t.dtype, in_t.device)
        if getattr(self, "_stage", None) is None or self._stage[0] != key:
            self._stage = (key, torch.empty(W * CH, dtype=in_t.dtype, device=in_t.device),
                           torch.empty(W * CH, dtype=in_t.dtype, device=in_t.device))
        _, s_buf, r_buf = self._stage
        for j in range(rounds):
            s_j = [min(max(m - j * CH, 0), CH) for m in send]
            r_j = [min(max(m - j * CH, 0), CH) for m in recv]
            o = 0
            for r in range(W):
                if s_j[r]:
                    s_buf[o:o + s_j[r]].copy_(in_t[pos[r] + j * CH: pos[r] + j * CH + s_j[r]])
                o += s_j[r]
            dist.all_to_all_single(r_buf[:sum(r_j)], s_buf[:sum(s_j)], r_j, s_j)
            o = 0
            for r in range(W):
                if r_j[r]:
                    out_t[roff[r] + j * CH: roff[r] + j * CH + r_j[r]].copy_(r_buf[o:o + r_j[r]])
                o += r_j[r]

    def exchange_and_merge(self, keys_t, counts_t, n):
        rk, rc, segs = self.exchange(keys_t, counts_t, n)
        return self.ops.merge_segments(rk, rc, segs)

    def verify_global(self, k, c, local_stream_sums):
        """Sum over ranks of the merged tables' checksums == sum over ranks of the streams' checksums
        (each a triple mod 2**64; carried as 32-bit halves so the reduction cannot overflow)."""
        got = self.ops.checksum(k, c)

        def halves(t):
            return [v & 0xFFFFFFFF for v in t] + [v >> 32 for v in t]
        dev = "cuda" if torch.cuda.is_available() and self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(halves(got) + halves(local_stream_sums), dtype=torch.int64, device=dev)
        self.dist.all_reduce(t)
        v = [int(x) for x in t.tolist()]
        if os.environ.get("ZOT_DEBUG"):
            import sys
            sys.stderr.write("verify_global: got %r want %r reduced %r\n" % (got, local_stream_sums, v))

        def join(lo, hi):
            return [(l + (h << 32)) & 0xFFFFFFFFFFFFFFFF for l, h in zip(lo, hi)]
        return join(v[0:3], v[3:6]) == join(v[6:9], v[9:12])

    def owned_slice(self, keys_t, n):
        """(start, end) of this rank's value range inside a sorted array every rank holds in full --
        how `zot dist` shards: both sets are cut with the same splitters, each rank runs zk_split on its
        slices, and split_counts() adds the three integers up."""
        pos = [0] + self.ops.lower_bound(keys_t, n, self.cuts) + [n]
        return pos[self.rank], pos[self.rank + 1]

    def split_counts(self, abc_local):
        """dist: all-reduce the (a, b, c) of the rank's key range."""
        dev = "cuda" if torch.cuda.is_available() and self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(list(abc_local), dtype=torch.int64, device=dev)
        self.dist.all_reduce(t)
        return tuple(int(x) for x in t.tolist())
