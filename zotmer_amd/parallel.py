"""
Multi-GPU kmerize / merge / dist: one process per GPU, launched by
`torch.distributed.run`; the "nccl" backend IS RCCL over xGMI on ROCm.

The reference is single-process; what is sharded here, and the ONE exchange step each command has:

  kmerize  reads are independent: every rank counts its own reads with no communication.  The per-rank
           sorted tables then meet in one all-to-all-v: the table is cut into `world` pieces by an owner
           function, piece r goes to rank r, and every rank union-sums the pieces it receives.
  merge    (commands/merge.py:127-163,165-253) rank r loads sets r, r+world, ... and k-way merges them
           locally; the result is cut and exchanged exactly like a kmerize table; hist and the
           count-weighted acgt are reduced over the ranks.
  dist     (commands/dist.py:94-168; library/dist.py:241-265) (a, b, c) is additive over any partition of the
           key space: both sets are cut by the same owner, every rank splits its piece, three integers are
           all-reduced.  Sets may arrive position-sharded (rank r holds the r-th contiguous chunk of each
           sorted file): the pieces a rank receives then concatenate to a sorted array, no merge needed.
  trim     any contiguous split, no collective.

Owner functions (both keep equal k-mers on one rank):
  "range"  contiguous value ranges with BALANCED cut points: every rank histograms its table over the top 16
           bits of the key (65 535 binary searches on the device, zk_lower_bound), the histograms are
           all-reduced, the bin that holds each of the world-1 quantiles is refined over the next 16 bits the
           same way, and the cuts are read off the summed prefix.  A sorted table is already partitioned by a
           monotone owner, so no data moves before the exchange, and the global sorted set is simply the
           concatenation of the ranks' pieces in rank order.
  "hash"   owner(x) = floor(murmer(x, seed) * world / 2^64) (basics.murmer, library/basics.py:191-229): balanced
           whatever the value distribution; the table is split by a stable device partition (zk_hash_partition)
           and the writer merges the ranks' (disjoint, sorted) pieces.

xGMI is a full mesh of point-to-point links, so the all-to-all is one hop per peer and per-link bound; payloads
are never ringed.  Two transports carry it: torch.distributed (`TorchComm`, also the "gloo" CPU path of the
tests) and the library's own RCCL seam zk_comm_* (`NativeComm`: grouped ncclSend/ncclRecv straight between the
tables, no staging copy).  The arithmetic on the data path is injected (`ops`), so the whole protocol runs
unchanged on CPU tensors under gloo in tests/test_parallel_gloo.py.
"""
import os

import numpy as np
import torch

M64 = 0xFFFFFFFFFFFFFFFF


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a process creates its first communicator; the commands' stdout is their
    result (`zot dist` prints a table).  File descriptor 1 points at stderr while communicators are made."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *a):
        import sys
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def splitters(K, world):
    """world-1 ascending EQUAL-WIDTH cut points of [0, 4**K) (static; the commands use balanced_cuts)."""
    space = 1 << (2 * K)
    return [(space * r) // world for r in range(1, world)]


def hash_owner(x, seed, world, murmer):
    """The hash-range owner of one k-mer (host restatement; the device form is zk_hash_partition)."""
    return (murmer(int(x), int(seed)) * world) >> 64


# ---------------------------------------------------------------------------------------------------
# transports
# ---------------------------------------------------------------------------------------------------
class TorchComm:
    """torch.distributed as the transport ("nccl" = RCCL on the GPU box, "gloo" on CPU tensors in the tests)."""

    name = "torch.distributed"
    # elements per peer and round.  One all_to_all_single with a per-peer message above 1 GiB arrived with its second
    # half wrong on this stack in round 1 (RCCL 2.26.6 / torch 2.10, one-rank self exchange: 2^27 int64 fine,
    # 2^27 + 1 corrupt; tests/test_gpu_exchange.py re-checks a bare contiguous tensor), so messages are cut to <= 256 MiB.
    CHUNK = 1 << 25

    def __init__(self, dist, device=None):
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        if device is None:
            device = "cuda" if (dist.get_backend() == "nccl" and torch.cuda.is_available()) else "cpu"
        self.device = device
        self._stage = None

    def all_reduce(self, vals, op="sum"):
        """Element-wise sum (mod 2^64) or max over the ranks of non-negative integers < 2^64.  A list gives a list of
        Python ints, a numpy uint64 array gives an array."""
        as_array = isinstance(vals, np.ndarray)
        a = np.ascontiguousarray(vals, dtype=np.uint64) if as_array else np.array([int(v) & M64 for v in vals], dtype=np.uint64)
        if self.world > 1 and a.size:
            lo = (a & np.uint64(0xFFFFFFFF)).astype(np.int64)
            hi = (a >> np.uint64(32)).astype(np.int64)
            if op == "max":
                th = torch.from_numpy(hi).to(self.device)
                tl = torch.from_numpy(lo).to(self.device)
                top = th.clone()
                self.dist.all_reduce(top, op=self.dist.ReduceOp.MAX)
                tl = torch.where(th == top, tl, torch.zeros_like(tl))
                self.dist.all_reduce(tl, op=self.dist.ReduceOp.MAX)
                a = (top.cpu().numpy().astype(np.uint64) << np.uint64(32)) | tl.cpu().numpy().astype(np.uint64)
            else:
                # 32-bit halves in int64 lanes: the sum over < 2^31 ranks cannot overflow
                t = torch.from_numpy(np.concatenate([lo, hi])).to(self.device)
                self.dist.all_reduce(t)
                r = t.cpu().numpy().astype(np.uint64)
                n = a.size
                with np.errstate(over="ignore"):
                    a = r[:n] + (r[n:] << np.uint64(32))
        return a if as_array else [int(v) for v in a]

    def all_gather_object(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def all_to_all_v(self, out_t, in_t, recv, send, send_off, recv_off):
        """out_t[recv_off[r] : +recv[r]] <- rank r's in_t[send_off[me] : +send[me]], in rounds of at most CHUNK elements
        per peer.  The slices of a round are not contiguous, so they go through two staging buffers."""
        dist, W, CH = self.dist, self.world, self.CHUNK
        contiguous = all(send_off[r + 1] == send_off[r] + send[r] for r in range(W - 1)) and \
            all(recv_off[r + 1] == recv_off[r] + recv[r] for r in range(W - 1))
        rounds = (max([0] + list(send) + list(recv)) + CH - 1) // CH
        rounds = self.all_reduce([rounds], "max")[0]
        if rounds <= 1 and contiguous:
            a, b = send_off[0], recv_off[0]
            dist.all_to_all_single(out_t[b:b + sum(recv)], in_t[a:a + sum(send)], list(recv), list(send))
            return
        key = (in_t.dtype, in_t.device)
        if self._stage is None or self._stage[0] != key or self._stage[1].numel() < W * CH:
            self._stage = (key, torch.empty(W * CH, dtype=in_t.dtype, device=in_t.device),
                           torch.empty(W * CH, dtype=in_t.dtype, device=in_t.device))
        _, s_buf, r_buf = self._stage
        for j in range(rounds):
            s_j = [min(max(m - j * CH, 0), CH) for m in send]
            r_j = [min(max(m - j * CH, 0), CH) for m in recv]
            o = 0
            for r in range(W):
                if s_j[r]:
                    s_buf[o:o + s_j[r]].copy_(in_t[send_off[r] + j * CH: send_off[r] + j * CH + s_j[r]])
                o += s_j[r]
            dist.all_to_all_single(r_buf[:sum(r_j)], s_buf[:sum(s_j)], r_j, s_j)
            o = 0
            for r in range(W):
                if r_j[r]:
                    out_t[recv_off[r] + j * CH: recv_off[r] + j * CH + r_j[r]].copy_(r_buf[o:o + r_j[r]])
                o += r_j[r]


class NativeComm:
    """The library's own RCCL seam (include/zotk.h zk_comm_*): grouped ncclSend / ncclRecv straight between the tables
    on the context's stream.  torch.distributed is used once, to carry the 128-byte communicator id to the ranks."""

    name = "zk_comm (RCCL send/recv)"

    def __init__(self, ctx, dist=None, world=None, rank=None, uid=None, self_loop=False):
        from zotmer_amd import native
        self.ctx = ctx
        # self_loop (tests on a box with one GPU): the piece a rank keeps goes through ncclSend / ncclRecv too, and a single
        # rank still calls ncclAllReduce -- zk_tune(ZK_TUNE_COMM_SELF_LOOP)
        self.self_loop = bool(self_loop)
        ctx.tune(comm_self_loop=int(self.self_loop))
        if dist is not None:
            world, rank = dist.get_world_size(), dist.get_rank()
            box = [native.Context.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            uid = box[0]
        elif uid is None:
            uid = native.Context.comm_unique_id()
        self.dist = dist
        self.world, self.rank = int(world), int(rank)
        ctx.comm_init(self.world, self.rank, uid)
        self.device = "cuda"

    def close(self):
        self.ctx.comm_destroy()
        if self.self_loop:
            self.ctx.tune(comm_self_loop=0)

    def all_reduce(self, vals, op="sum"):
        as_array = isinstance(vals, np.ndarray)
        a = np.ascontiguousarray(vals, dtype=np.uint64) if as_array else np.array([int(v) & M64 for v in vals], dtype=np.uint64)
        if (self.world > 1 or self.self_loop) and a.size:
            a = self.ctx.allreduce_u64(a, 0 if op == "sum" else 1)
        return a if as_array else [int(v) for v in a]

    def all_gather_object(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def barrier(self):
        self.ctx.sync()
        if self.world > 1 and self.dist is not None:
            self.dist.barrier()

    def all_to_all_v(self, out_t, in_t, recv, send, send_off, recv_off):
        self.ctx.all_to_all_v(in_t.data_ptr(), send_off[:self.world], send, out_t.data_ptr(), recv_off[:self.world], recv,
                              in_t.element_size())
        self.ctx.sync()


# ---------------------------------------------------------------------------------------------------
# data-path arithmetic on the GPU (the tests substitute the CPU oracle on CPU tensors)
# ---------------------------------------------------------------------------------------------------
class GpuOps:
    """Data-path operations on torch CUDA tensors through libzotk (the tensors only carry bytes: uint64 k-mers travel
    as int64, uint32 counts as int32, uint64 counts as int64)."""

    def __init__(self, ctx):
        self.ctx = ctx
        from zotmer_amd import native
        self._borrow = native.DeviceArray.borrow
        self._mk = self._mc = None

    def empty(self, n, dtype):
        return torch.empty(max(int(n), 1), dtype=dtype, device="cuda")

    def k_array(self, t, n, off=0):
        return self._borrow(self.ctx, t.data_ptr() + 8 * off, np.uint64, n, keep=t)

    def c_array(self, t, n, off=0):
        if t.dtype == torch.int64:
            return self._borrow(self.ctx, t.data_ptr() + 8 * off, np.uint64, n, keep=t)
        return self._borrow(self.ctx, t.data_ptr() + 4 * off, np.uint32, n, keep=t)

    def lower_bound(self, keys_t, n, cuts):
        return self.ctx.lower_bound(self.k_array(keys_t, n), cuts)

    def hash_partition(self, keys_t, counts_t, n, world, seed):
        """-> (keys tensor, counts tensor | None, offsets[world + 1]): the table split by hash owner, pieces still sorted"""
        ok_t = self.empty(n, torch.int64)
        oc_t = self.empty(n, counts_t.dtype) if counts_t is not None else None
        _, _, offs = self.ctx.hash_partition(self.k_array(keys_t, n), self.c_array(counts_t, n) if counts_t is not None else None,
                                             world, seed, out=(self.k_array(ok_t, n), self.c_array(oc_t, n) if oc_t is not None else None))
        return ok_t, oc_t, offs

    def before_comm(self):
        self.ctx.sync()                 # our kernels run on the ctx's own stream

    def after_comm(self):
        torch.cuda.synchronize()

    def merge_segments(self, keys_t, counts_t, segs, want_acgt=False):
        """k-way union-sum of the sorted segments [(offset, length)] of the receive buffers (zk_merge_n: a tree of
        merge-path passes inside the library's workspace; the output buffers are kept between steps, so a steady-state
        step allocates nothing)."""
        parts = [(self.k_array(keys_t, n, o), self.c_array(counts_t, n, o)) for o, n in segs if n]
        if not parts:
            r = (self.k_array(keys_t, 0), self.c_array(counts_t, 0))
            return r + ([0, 0, 0, 0],) if want_acgt else r
        total = sum(p[0].n for p in parts)
        cdt = parts[0][1].dtype
        if len(parts) == 1 and not want_acgt:
            return parts[0]
        if self._mk is None or self._mk.n < total or self._mc.dtype != cdt:
            self._mk = self._mc = None
            self._mk = self.ctx.empty(total + total // 16, np.uint64)
            self._mc = self.ctx.empty(total + total // 16, cdt)
        k, c, acgt = self.ctx.merge_n(parts, out=(self._mk, self._mc))
        return (k, c, acgt) if want_acgt else (k, c)

    def mirror_expand(self, k, c, K):
        """the owned canonical piece -> (kmers, counts) of both strands (zk_mirror_expand; output buffers kept between steps)"""
        if k.n == 0:
            return k, c
        if getattr(self, "_xk", None) is None or self._xk.n < 2 * k.n:
            self._xk = self._xc = None
            self._xk = self.ctx.empty(2 * k.n + k.n // 8, np.uint64)
            self._xc = self.ctx.empty(2 * k.n + k.n // 8, np.uint32)
        return self.ctx.mirror_expand(k, c, K, out=(self._xk, self._xc))

    def dedupe(self, keys_t, n, shift=0):
        """Measure.prep (commands/dist.py:43-49): x >> shift, adjacent duplicates dropped -> (tensor, n)"""
        out_t = self.empty(n, torch.int64)
        from zotmer_amd import native
        import ctypes as C
        m = C.c_uint64(0)
        self.ctx._check(self.ctx.lib.zk_project_dedupe(self.ctx.h, keys_t.data_ptr(), int(n), int(shift), out_t.data_ptr(), int(n), C.byref(m)))
        return out_t, m.value

    def split(self, x_t, nx, y_t, ny):
        return self.ctx.split(self.k_array(x_t, nx), self.k_array(y_t, ny))

    def checksum(self, k, c):
        return self.ctx.checksum_counts(k, c)

    def hist(self, c):
        return self.ctx.hist(c)

    def to_tensors(self, k, c):
        """library arrays -> torch tensors (a copy: the library's buffers are reused by the next step)"""
        kt = self.empty(k.n, torch.int64)
        ct = self.empty(c.n, torch.int64 if c.dtype.itemsize == 8 else torch.int32)
        if k.n:
            self.ctx._check(self.ctx.lib.zk_copy(self.ctx.h, kt.data_ptr(), k.ptr, k.nbytes))
            self.ctx._check(self.ctx.lib.zk_copy(self.ctx.h, ct.data_ptr(), c.ptr, c.nbytes))
            self.ctx.sync()
        return kt, ct, k.n


# ---------------------------------------------------------------------------------------------------
# the protocol
# ---------------------------------------------------------------------------------------------------
class Exchange:
    """Partition by owner, all-to-all-v, merge: the one communication step of kmerize / merge / dist."""

    HIST_BITS = 16          # histogram resolution per level of the balanced splitters (2 levels)

    def __init__(self, ctx, dist, K, ops=None, owner="range", seed=0, comm=None):
        assert owner in ("range", "hash")
        self.K, self.owner, self.seed = K, owner, seed
        self.comm = comm if comm is not None else TorchComm(dist)
        self.dist = dist
        self.world, self.rank = self.comm.world, self.comm.rank
        self.ops = ops if ops is not None else GpuOps(ctx)
        self.cuts = splitters(K, self.world)      # static cuts until balanced_cuts() is called
        self._rk = self._rc = None                 # receive buffers, kept between steps (grown on demand)
        self.by_canonical = False                  # pieces own k-mers by their CANONICAL form (kmerize_finish): not value ranges

    # back-compatible knob used by the tests to force several rounds per all-to-all
    @property
    def CHUNK(self):
        return getattr(self.comm, "CHUNK", None)

    @CHUNK.setter
    def CHUNK(self, v):
        self.comm.CHUNK = v

    # ---- balanced value-range splitters ---------------------------------------------------------------
    def balanced_cuts(self, tables, key_bits=None):
        """world-1 cut values that give every rank (nearly) the same number of entries, counted over ALL ranks' tables.
        tables: [(keys tensor, n)] held by this rank.  Two levels of 2^16-bin histograms (module docstring)."""
        W, ops = self.world, self.ops
        kb = key_bits if key_bits is not None else 2 * self.K
        if W == 1:
            self.cuts = []
            return self.cuts
        b1 = min(self.HIST_BITS, kb)
        s1 = kb - b1

        def cumulative(queries):
            tot = np.zeros(len(queries), dtype=np.uint64)
            for t, n in tables:
                if n:
                    tot += np.asarray(ops.lower_bound(t, n, queries), dtype=np.uint64)
            return tot

        def u64(v):
            return np.array([v], dtype=np.uint64)

        q1 = np.arange(1, 1 << b1, dtype=np.uint64) << np.uint64(s1)
        n_local = sum(n for _, n in tables)
        red = self.comm.all_reduce(np.concatenate([cumulative(q1), u64(n_local)]))
        total = int(red[-1])
        cum1 = np.concatenate([u64(0), red[:-1]])                  # cum1[b] = entries below bin b's lower edge
        targets = [(total * r) // W for r in range(1, W)]
        # the bin that holds each target quantile: the last b with cum1[b] <= target
        bins = [int(np.searchsorted(cum1, np.uint64(tg), side="right")) - 1 for tg in targets]
        b2 = min(self.HIST_BITS, s1)
        if b2 == 0:
            self.cuts = [b << s1 for b in bins]
        else:
            s2 = s1 - b2
            inner = np.arange(1, 1 << b2, dtype=np.uint64) << np.uint64(s2)
            q2 = np.concatenate([np.uint64(b << s1) + inner for b in bins])
            red2 = self.comm.all_reduce(cumulative(q2))
            per = (1 << b2) - 1
            cuts = []
            for i, (b, tg) in enumerate(zip(bins, targets)):
                cum2 = np.concatenate([cum1[b:b + 1], red2[i * per:(i + 1) * per]])
                j = int(np.searchsorted(cum2, np.uint64(tg), side="right")) - 1
                cuts.append((b << s1) + (j << s2))
            self.cuts = cuts
        for a, b in zip(self.cuts, self.cuts[1:]):
            assert a <= b
        return self.cuts

    # ---- the exchange -------------------------------------------------------------------------------
    def _partition(self, keys_t, counts_t, n):
        """-> (keys tensor, counts tensor, piece boundaries pos[world + 1])"""
        if self.owner == "hash":
            return self.ops.hash_partition(keys_t, counts_t, n, self.world, self.seed)
        pos = [0] + [int(p) for p in (self.ops.lower_bound(keys_t, n, self.cuts) if self.cuts else [])] + [n]
        return keys_t, counts_t, pos

    def exchange(self, keys_t, counts_t, n):
        """keys_t (int64) / counts_t (int32 | int64 | None): this rank's sorted table in its first n entries.
        Returns (recv keys, recv counts, [(offset, length) per source rank])."""
        ops, W = self.ops, self.world
        keys_t, counts_t, pos = self._partition(keys_t, counts_t, n)
        send = [pos[r + 1] - pos[r] for r in range(W)]
        ops.before_comm()
        recv = self._exchange_sizes(send)
        total = sum(recv)
        cdt = counts_t.dtype if counts_t is not None else None
        if self._rk is None or self._rk.numel() < total or (cdt is not None and (self._rc is None or self._rc.dtype != cdt)):
            self._rk = self._rc = None
            self._rk = ops.empty(total + total // 16, torch.int64)
            self._rc = ops.empty(total + total // 16, cdt) if cdt is not None else None
        rk, rc = self._rk, self._rc
        roff = [0]
        for m in recv:
            roff.append(roff[-1] + m)
        self.comm.all_to_all_v(rk, keys_t, recv, send, pos, roff)
        if counts_t is not None:
            self.comm.all_to_all_v(rc, counts_t, recv, send, pos, roff)
        ops.after_comm()
        return rk, (rc if counts_t is not None else None), [(roff[r], recv[r]) for r in range(W)]

    def _exchange_sizes(self, send):
        """recv[r] = what rank r sends to me: an all-to-all of one integer per peer, done as an all-reduce of the
        world x world size matrix (tiny; works on every transport)."""
        W = self.world
        if W == 1:
            return list(send)
        mat = [0] * (W * W)
        for r in range(W):
            mat[self.rank * W + r] = send[r]
        mat = self.comm.all_reduce(mat)
        return [mat[r * W + self.rank] for r in range(W)]

    def exchange_and_merge(self, keys_t, counts_t, n, want_acgt=False):
        rk, rc, segs = self.exchange(keys_t, counts_t, n)
        return self.ops.merge_segments(rk, rc, segs, want_acgt) if want_acgt else self.ops.merge_segments(rk, rc, segs)

    # ---- reductions -----------------------------------------------------------------------------------
    def verify_global(self, k, c, local_input_sums):
        """Sum over ranks of the owned pieces' checksums == sum over ranks of the inputs' checksums (triples mod 2^64)."""
        got = self.ops.checksum(k, c)
        v = self.comm.all_reduce(list(got) + list(local_input_sums))
        if os.environ.get("ZOT_DEBUG"):
            import sys
            sys.stderr.write("verify_global: got %r want %r reduced %r\n" % (got, local_input_sums, v))
        return v[0:3] == v[3:6]

    def split_counts(self, abc_local):
        """dist: all-reduce the (a, b, c) of the rank's piece."""
        return tuple(self.comm.all_reduce(list(abc_local)))

    def owned_slice(self, keys_t, n):
        """(start, end) of this rank's value range inside a sorted array every rank holds in full (range owner)."""
        pos = [0] + [int(p) for p in self.ops.lower_bound(keys_t, n, self.cuts)] + [n]
        return pos[self.rank], pos[self.rank + 1]

    # ---- product functions ----------------------------------------------------------------------------
    def kmerize_finish(self, keys_t, counts_t, n):
        """`zot kmerize` over the ranks: keys_t / counts_t = this rank's counted CANONICAL list (zk_kmerize with
        ZK_KMERIZE_CANONICAL_ONLY: half the size of the both-strand table).  The canonical lists are cut by owner, exchanged
        and union-summed; each rank then rebuilds both strands of the k-mers whose canonical form it owns
        (zk_mirror_expand).  The owned pieces partition the global table (x and rc x always live on the same rank) but are
        not value ranges: the writer merges them (gather_to_root).  Returns (kmers, counts) of both strands."""
        self.by_canonical = True
        if self.owner == "range":
            self.balanced_cuts([(keys_t, n)])
        k, c = self.exchange_and_merge(keys_t, counts_t, n)
        return self.ops.mirror_expand(k, c, self.K)

    def merge_sets(self, keys_t, counts_t, n):
        """`zot merge` over the ranks (commands/merge.py:127-163,165-253).  keys_t / counts_t (int64, 64-bit counts): the
        k-way merge of the sets THIS rank loaded (sets r, r+world, ...).  Returns dict(k, c, acgt, hist, n_global):
        k / c = the piece this rank owns, acgt / hist / n_global reduced over all ranks."""
        if self.owner == "range":
            self.balanced_cuts([(keys_t, n)])
        k, c, acgt = self.exchange_and_merge(keys_t, counts_t, n, want_acgt=True)
        acgt = self.comm.all_reduce(acgt)
        hist = {}
        for h in self.comm.all_gather_object(self.ops.hist(c)):
            for v, f in h.items():
                hist[v] = hist.get(v, 0) + f
        n_global = self.comm.all_reduce([len_of(k)])[0]
        return dict(k=k, c=c, acgt=acgt, hist=hist, n_global=n_global)

    def dist_pair(self, x_t, nx, y_t, ny, shift=0):
        """`zot dist` for one pair (commands/dist.py:43-49,145-159; library/dist.py:241-265).  x_t / y_t: this rank's
        contiguous chunk of each sorted set (position-sharded; rank order = value order), or the whole set on one
        rank and nothing on the others.  shift = 2 * (fK - K) projects to the K-prefix.  Returns the global
        (a, b, c) and (|X|, |Y|) after projection."""
        ops = self.ops
        if shift:
            x_t, nx = ops.dedupe(x_t, nx, shift)
            y_t, ny = ops.dedupe(y_t, ny, shift)
        if self.world > 1:
            if self.owner == "range":
                self.balanced_cuts([(x_t, nx), (y_t, ny)], key_bits=2 * self.K)
            sides = []
            for t, n in ((x_t, nx), (y_t, ny)):
                rk, _, segs = self.exchange(t, None, n)
                m = sum(s[1] for s in segs)
                # pieces arrive in source-rank order = value order: the concatenation is sorted; a projected key cut
                # by a chunk boundary shows up twice in a row, so dedupe once more
                piece, m = ops.dedupe(rk, m, 0)
                sides.append((piece, m))
            (x_t, nx), (y_t, ny) = sides
        abc = ops.split(x_t, nx, y_t, ny)
        abc = self.split_counts(abc)
        return abc, (abc[0] + abc[1], abc[0] + abc[2])

    def gather_to_root(self, k, c, root=0):
        """The ranks' owned pieces on `root` as one sorted table (the writer of the single output file): rank order for the
        range owner; for the hash owner the pieces are disjoint but interleaved, so the root merges them (the
        'streaming W-way merge on write' of SURVEY 8(e)).  Returns (k, c) on root, (None, None) elsewhere."""
        ops, W = self.ops, self.world
        if W == 1:
            return k, c
        kt, ct, n = ops.to_tensors(k, c)
        send = [0] * W
        send[root] = n
        pos = [0] * (W + 1)
        for r in range(root + 1, W + 1):
            pos[r] = n
        ops.before_comm()
        recv = self._exchange_sizes(send)
        total = sum(recv)
        rk = ops.empty(total, torch.int64)
        rc = ops.empty(total, ct.dtype)
        roff = [0]
        for m in recv:
            roff.append(roff[-1] + m)
        self.comm.all_to_all_v(rk, kt, recv, send, pos, roff)
        self.comm.all_to_all_v(rc, ct, recv, send, pos, roff)
        ops.after_comm()
        if self.rank != root:
            return None, None
        segs = [(roff[r], recv[r]) for r in range(W)]
        if self.owner == "hash" or self.by_canonical:
            return ops.merge_segments(rk, rc, segs)
        return ops.merge_segments(rk, rc, [(0, total)])


def len_of(a):
    return a.n if hasattr(a, "n") else len(a)


# the name round 1 used (value-range owner, static cuts unless balanced_cuts is called)
RangeExchange = Exchange


def init_from_env(ctx=None, owner=None):
    """(dist, Exchange factory) for a process launched by torch.distributed.run: reads RANK / WORLD_SIZE / LOCAL_RANK and
    MASTER_*; returns None when WORLD_SIZE <= 1.  ZOT_COMM=native|torch picks the transport (default: native on
    the GPU), ZOT_OWNER=range|hash the owner function."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not dist.is_initialized():
        torch.cuda.set_device(local)
        with stdout_to_stderr():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            dist.barrier()          # makes torch's communicator (and RCCL's banner) now
    return dist


def make_comm(ctx, dist, notes=None):
    """The transport ZOT_COMM asks for: 'native' (zk_comm_*, the default on the GPU) or 'torch'.  The native transport is
    probed once -- a small all-to-all-v and an all-reduce checked on every rank -- and every rank learns through
    torch.distributed whether ALL of them passed; if not, all of them use torch.distributed instead, and say so (`notes`, a
    list, receives the reason: callers print it / put it in their report -- the switch is never silent)."""
    want = os.environ.get("ZOT_COMM", "native")
    if not (want == "native" and dist.get_backend() == "nccl"):
        return TorchComm(dist)
    comm, why = None, ""
    try:
        with stdout_to_stderr():
            comm = NativeComm(ctx, dist)
        W, r = comm.world, comm.rank
        src = torch.arange(W * 4, dtype=torch.int64, device="cuda") + 1000 * r
        dst = torch.zeros(W * 4, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        comm.all_to_all_v(dst, src, [4] * W, [4] * W, [4 * i for i in range(W + 1)], [4 * i for i in range(W + 1)])
        want_t = torch.cat([torch.arange(4 * r, 4 * r + 4, dtype=torch.int64, device="cuda") + 1000 * q for q in range(W)])
        ok = bool(torch.equal(dst, want_t)) and comm.all_reduce([r + 1, 1 << 40]) == [W * (W + 1) // 2, (W << 40) & M64]
        if ok:
            # ragged pieces, empty ones among them, in rounds of 16 bytes: the paths a real exchange takes (peers that send
            # nothing, a last partial round) and the uniform probe above does not
            cnt = lambda a, b: (7 * a + 3 * b) % 6                                      # noqa: E731 -- elements rank a sends to rank b
            send, recv = [cnt(r, q) for q in range(W)], [cnt(q, r) for q in range(W)]
            so, ro = [0] + list(np.cumsum(send)), [0] + list(np.cumsum(recv))
            src = torch.cat([torch.arange(send[q], dtype=torch.int64, device="cuda") + 100 * q + 10000 * r for q in range(W)] +
                            [torch.zeros(1, dtype=torch.int64, device="cuda")])
            dst = torch.full((int(ro[-1]) + 1,), -1, dtype=torch.int64, device="cuda")
            want_t = torch.cat([torch.arange(recv[q], dtype=torch.int64, device="cuda") + 100 * r + 10000 * q for q in range(W)] +
                               [torch.full((1,), -1, dtype=torch.int64, device="cuda")])
            torch.cuda.synchronize()
            ctx.tune(comm_chunk=16)
            try:
                comm.all_to_all_v(dst, src, recv, send, [int(v) for v in so], [int(v) for v in ro])
            finally:
                ctx.tune(comm_chunk=0)
            ok = bool(torch.equal(dst, want_t))
        if not ok:
            why = "probe exchange returned wrong data"
    except Exception as e:       # noqa: BLE001 -- any failure of the optional transport selects the other one, on every rank
        ok, why = False, "%s: %s" % (type(e).__name__, e)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cuda")
    if dist.get_world_size() > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return comm
    if comm is not None:
        try:
            comm.close()
        except Exception:        # noqa: BLE001
            pass
    msg = "zk_comm transport not usable on every rank (%s); torch.distributed transport used" % (why or "another rank failed")
    if notes is not None:
        notes.append(msg)
    import sys
    sys.stderr.write(msg + "\n")
    return TorchComm(dist)
