"""Run by tests/test_gpu_multigpu.py in its own process (torch first, as the commands do under torch.distributed.run).

BASELINE config 4 in miniature on ONE device: 64 synthetic sets (zotmer_amd/synth.py config 4 generator, scaled down)
merged over 8 LOGICAL ranks -- eight threads, each with its own zk_ctx on the same GPU, running the PRODUCT functions of
zotmer_amd/parallel.py (Exchange.merge_sets / dist_pair / gather_to_root with GpuOps, i.e. zk_merge_n, zk_lower_bound,
zk_hash_partition, zk_project_dedupe, zk_split on the device).  Only the transport is a stand-in: `ThreadComm` moves the
pieces with device-to-device copies and reduces integers under a barrier, where the real thing calls RCCL.  Compared
bit for bit with one oracle merge of all 64 sets, for both owner functions; dist at K = 25 and projected to K = 12; the kmerize
exchange (canonical lists cut by owner, exchanged, strands rebuilt by the owner) at K = 25 and at K = 31.
"""
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zkoracle as zo                 # noqa: E402
from zotmer_amd import native, parallel, synth    # noqa: E402

# Eight contexts share ONE GPU here, which no product run does (one process and one context per GPU).  The radix-sort pipelines are
# persistent kernels that take whole CUs (150 KB of LDS a workgroup) and whose scanner workgroups wait for the counts of 32 tiles
# at a time: eight of them at once can leave every one with too few resident workgroups to finish a batch (measured: 2 of 30 runs
# of this script hit the bounded spin, also on the round-3 tree).  What this script tests is the exchange logic over 8 ranks, not
# eight kernels sharing a card, so the library calls of the eight threads are serialised: one context on the GPU at a time.
_device_lock = threading.Lock()


def _serialise_library_calls():
    lib = native.load()
    for name in native.SIGNATURES:
        fn = getattr(lib, name)

        def locked(*a, _fn=fn):
            with _device_lock:
                return _fn(*a)
        setattr(lib, name, locked)


W = 8
SCALE = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0004        # 64 sets x 20 000 keys from a pool of 80 000
K = 25
K31 = 31


class ThreadComm:
    """The transport of parallel.Exchange for W threads of one process."""
    name = "threads"
    shared = None

    def __init__(self, rank):
        self.world, self.rank = W, rank
        self.device = "cuda"

    @classmethod
    def setup(cls):
        cls.shared = dict(bar=threading.Barrier(W), slots=[None] * W, err=[])

    def _gather(self, obj):
        s = self.shared
        s["slots"][self.rank] = obj
        s["bar"].wait()
        out = list(s["slots"])
        s["bar"].wait()
        return out

    def all_reduce(self, vals, op="sum"):
        as_array = isinstance(vals, np.ndarray)
        a = np.ascontiguousarray(vals, dtype=np.uint64) if as_array else np.array([int(v) & parallel.M64 for v in vals], dtype=np.uint64)
        parts = self._gather(a)
        with np.errstate(over="ignore"):
            r = parts[0].copy()
            for p in parts[1:]:
                r = np.maximum(r, p) if op == "max" else r + p
        return r if as_array else [int(v) for v in r]

    def all_gather_object(self, obj):
        return self._gather(obj)

    def barrier(self):
        self.shared["bar"].wait()

    def all_to_all_v(self, out_t, in_t, recv, send, send_off, recv_off):
        torch.cuda.synchronize()
        views = self._gather((in_t, list(send), list(send_off)))
        for src in range(W):
            t, s_cnt, s_off = views[src]
            n = s_cnt[self.rank]
            assert n == recv[src]
            if n:
                out_t[recv_off[src]:recv_off[src] + n].copy_(t[s_off[self.rank]:s_off[self.rank] + n])
        torch.cuda.synchronize()
        self.shared["bar"].wait()


def make_sets():
    sets = []
    for s in range(64):
        a = synth.config4_set_args(s, SCALE)
        k = synth.set_keys(a["seed"], a["first"], a["count"], a["key_bits"], mul=a["mul"], add=a["add"], mod=a["mod"])
        sets.append((k, synth.set_counts(a["seed"], k)))
    return sets


def worker(rank, owner, host_sets, results):
    try:
        ctx = native.Context(0)
        comm = ThreadComm(rank)
        ex = parallel.Exchange(ctx, None, K, owner=owner, seed=11, comm=comm)
        # the sets of this rank, generated ON THE DEVICE (zk_synth_keys / zk_synth_counts) and checked against synth.py
        mine = []
        for s in range(rank, 64, W):
            a = synth.config4_set_args(s, SCALE)
            k, c = ctx.synth_set(a["seed"], a["first"], a["count"], a["key_bits"], mul=a["mul"], add=a["add"], mod=a["mod"])
            assert np.array_equal(k.to_host(), host_sets[s][0]) and np.array_equal(c.to_host(), host_sets[s][1]), "device generator"
            mine.append((k, c))
        k, c, _ = ctx.merge_n(mine)                                    # 8 sets per rank
        kt, ct, n = ex.ops.to_tensors(k, c)
        sums_in = [0, 0, 0]
        for sk, sc in mine:
            for i, v in enumerate(ctx.checksum_counts(sk, sc)):
                sums_in[i] = (sums_in[i] + v) & parallel.M64
        res = ex.merge_sets(kt, ct, n)
        assert ex.verify_global(res["k"], res["c"], sums_in), "checksum of checksums"
        owned = res["k"].n
        gk, gc = ex.gather_to_root(res["k"], res["c"])
        out = dict(owned=owned, acgt=res["acgt"], hist=res["hist"], n_global=res["n_global"])
        if rank == 0:
            out["k"], out["c"] = gk.to_host(), gc.to_host()
        # dist: sets 0 and 1, position-sharded over the ranks, at K = 25 and projected to K = 12
        for kk in (25, 12):
            exd = parallel.Exchange(ctx, None, kk, owner=owner, seed=11, comm=comm)
            pieces = []
            for s in (0, 1):
                ch = np.array_split(host_sets[s][0], W)[rank]
                pieces.append((torch.from_numpy(ch.view(np.int64).copy()).cuda() if len(ch) else torch.empty(1, dtype=torch.int64, device="cuda"), len(ch)))
            abc, sizes = exd.dist_pair(pieces[0][0], pieces[0][1], pieces[1][0], pieces[1][1], shift=2 * (K - kk))
            out["abc%d" % kk], out["sz%d" % kk] = abc, sizes
        # kmerize: every rank counts its own reads as a CANONICAL list, the lists are exchanged, the owner rebuilds the strands
        R = 2500
        kw = dict(genome=40000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.0005))
        d = ctx.synth_reads(synth.DEFAULT_SEED, rank * R, R, 150, **kw)
        ck, cc, st = ctx.kmerize(d, K, native.KMERIZE_CANONICAL_ONLY)
        exk = parallel.Exchange(ctx, None, K, owner=owner, seed=11, comm=comm)
        kt, ct, n = exk.ops.to_tensors(ck, cc)
        bk, bc = exk.kmerize_finish(kt, ct, n)
        assert exk.verify_global(bk, bc, ctx.stream_checksum(d, K)), "kmerize: checksum of checksums"
        out["kz_owned"] = bk.n
        gk, gc = exk.gather_to_root(bk, bc)
        if rank == 0:
            out["kz_k"], out["kz_c"] = gk.to_host(), gc.to_host()
        # the same at K = 31 (BASELINE config 5: 62-bit keys -- the balanced cuts' 16 + 16-bit histograms sit at the top of a key
        # twelve bits wider, the owner of a k-mer is the owner of its canonical form)
        ck, cc, st = ctx.kmerize(d, K31, native.KMERIZE_CANONICAL_ONLY)
        exk = parallel.Exchange(ctx, None, K31, owner=owner, seed=11, comm=comm)
        kt, ct, n = exk.ops.to_tensors(ck, cc)
        bk, bc = exk.kmerize_finish(kt, ct, n)
        assert exk.verify_global(bk, bc, ctx.stream_checksum(d, K31)), "kmerize K = 31: checksum of checksums"
        out["kz31_owned"] = bk.n
        gk, gc = exk.gather_to_root(bk, bc)
        if rank == 0:
            out["kz31_k"], out["kz31_c"] = gk.to_host(), gc.to_host()
        results[rank] = out
        ctx.close()
    except BaseException as e:          # noqa: BLE001
        import traceback
        ThreadComm.shared["err"].append("rank %d: %s" % (rank, traceback.format_exc()))
        try:
            ThreadComm.shared["bar"].abort()
        except Exception:
            pass
        raise e


def main():
    _serialise_library_calls()
    host_sets = make_sets()
    zs, zc, zacgt = zo.merge_n(K, host_sets)
    hv, hf = zo.hist(zc)
    for owner in ("range", "hash"):
        ThreadComm.setup()
        results = [None] * W
        th = [threading.Thread(target=worker, args=(r, owner, host_sets, results)) for r in range(W)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        errs = ThreadComm.shared["err"]
        if errs:
            root = [e for e in errs if "BrokenBarrierError" not in e] or errs
            print("LOGICAL-RANKS-FAILED", owner, "\n" + "\n".join(root[:2]), flush=True)
            sys.exit(1)
        r0 = results[0]
        assert np.array_equal(r0["k"], zs), owner + ": merged k-mers differ from the oracle"
        assert np.array_equal(r0["c"], zc), owner + ": merged counts differ from the oracle"
        sizes = [r["owned"] for r in results]
        assert sum(sizes) == len(zs) and max(sizes) <= 1.2 * len(zs) / W, (owner, sizes)
        for r in results:
            assert [int(v) for v in r["acgt"]] == [int(v) for v in zacgt]
            assert r["hist"] == {int(a): int(b) for a, b in zip(hv, hf)}
            assert r["n_global"] == len(zs)
            for kk in (25, 12):
                pa, pb = zo.project_dedupe(host_sets[0][0], 2 * (K - kk)), zo.project_dedupe(host_sets[1][0], 2 * (K - kk))
                assert tuple(int(v) for v in r["abc%d" % kk]) == zo.split(pa, pb), (owner, kk)
                assert tuple(int(v) for v in r["sz%d" % kk]) == (len(pa), len(pb))
        R = 2500
        kw = dict(genome=40000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.0005))
        reads = []
        for r in range(W):
            reads += synth.read_strings(synth.DEFAULT_SEED, r * R, R, 150, **kw)
        wantk = zo.kmerize(K, reads)
        assert np.array_equal(r0["kz_k"], wantk["kmers"]) and np.array_equal(r0["kz_c"], wantk["counts"]), owner + ": kmerize over 8 ranks"
        ksz = [r["kz_owned"] for r in results]
        assert sum(ksz) == len(wantk["kmers"]) and max(ksz) <= 1.25 * len(wantk["kmers"]) / W, (owner, ksz)
        want31 = zo.kmerize(K31, reads)
        assert np.array_equal(r0["kz31_k"], want31["kmers"]) and np.array_equal(r0["kz31_c"], want31["counts"]), owner + ": kmerize K = 31 over 8 ranks"
        ksz31 = [r["kz31_owned"] for r in results]
        assert sum(ksz31) == len(want31["kmers"]) and max(ksz31) <= 1.25 * len(want31["kmers"]) / W, (owner, ksz31)
        print("LOGICAL-RANKS-OK", owner, len(zs), sizes, ksz, ksz31)


if __name__ == "__main__":
    main()
