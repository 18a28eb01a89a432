"""
GPU orchestration shared by the commands: one lazily created zk_ctx, and the batch accumulator that
plays the role of the reference's KmerAccumulator2 + spill files (zotmer/commands/kmerize.py:370-437,
528-553): every batch of reads is counted on the device (zk_kmerize) and union-summed into a resident
sorted table (zk_union_sum); nothing is spilled to disk -- the final arrays do not depend on where
the batches are cut (verified against the reference's -m 1 run, tests/golden).
"""
import os

import numpy as np

from zotmer_amd import native

_ctx = None


def context():
    """The process-wide device context.  ZOT_DEVICE selects the GPU; under torch.distributed.run it is LOCAL_RANK (one
    process per GPU).  Raises if there is none: this build has no CPU path."""
    global _ctx
    if _ctx is None:
        dev = os.environ.get("ZOT_DEVICE")
        if dev is None:
            dev = os.environ.get("LOCAL_RANK", "0") if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "0"
        _ctx = native.Context(int(dev))
    return _ctx


def distributed():
    """(dist, world, rank) when launched by torch.distributed.run with WORLD_SIZE > 1, else (None, 1, 0).  torch is
    imported (before libzotk is loaded, so that both bind the same HIP runtime) only in that case."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, 1, 0
    from zotmer_amd import parallel
    dist = parallel.init_from_env()
    return dist, dist.get_world_size(), dist.get_rank()


def close():
    global _ctx
    if _ctx is not None:
        _ctx.close()
        _ctx = None


def batch_bytes_for(ctx, requested=None):
    """How much base stream to count per device batch.  zk_kmerize needs about 16 B of sort buffer per stream byte, up
    to 24 B per distinct canonical k-mer for the mirror step and 12 B per output entry; 80 B per byte is a safe
    envelope.  Half of the free memory is left for the table that grows while batches are counted (KmerTable checks
    again before every batch)."""
    free, _ = ctx.mem_info()
    b = max(1 << 20, int(free * 0.5) // 80)
    if requested:
        b = min(b, int(requested))
    return min(b, 6 << 30)


class KmerTable:
    """Sorted distinct k-mers + counts on the device, grown batch by batch.

    Every batch is counted into its own sorted table; tables are union-summed pairwise like a binary counter (two
    tables of the same level make one of the next), so n batches cost O(n log n) table traffic instead of the
    O(n^2) of re-merging one resident table per batch, and nothing is copied per batch but the batch's own result.
    The final arrays do not depend on where the batches are cut (the reference's -m 1 run, tests/golden)."""

    def __init__(self, ctx, K, subsample=None, baits=None):
        self.ctx, self.K = ctx, K
        self.flags = native.KMERIZE_CANONICAL
        self.p, self.seed = 0.0, 0
        if subsample is not None:
            self.flags |= native.KMERIZE_SUBSAMPLE
            self.p, self.seed = subsample
            baits = None              # `if d is not None: ... elif B is not None:` (kmerize.py:494-520): -D wins over -C
        self.baits = baits            # DeviceArray of sorted both-strand bait k-mers, or None
        self.parts = []               # [(level, kmers, counts)], levels strictly decreasing towards the end
        self.acgt = [0, 0, 0, 0]
        self.instances = 0
        self._out = None              # the batch output buffers, reused while they are big enough

    def add_fastq_text(self, text, line_phase):
        """Count one batch of raw FASTQ text (whole lines): parsed on the device (zk_fastq_mask)."""
        self._make_room(2 * len(text))
        d = self.ctx.upload_stream(text)
        if d.n == 0:
            return
        stream, _ = self.ctx.fastq_mask(d, line_phase)
        del d
        self.add_device_stream(stream)

    def add_stream(self, stream_host):
        """Count one batch (uint8 base stream on the host) into the table."""
        self._make_room(len(stream_host))
        self.add_device_stream(self.ctx.upload_stream(stream_host))

    def _make_room(self, n_bytes):
        """Before a batch: if the free memory no longer covers the batch's envelope, fold the waiting tables into one
        (frees their slack and the merge outputs' upper-bound padding); fail clearly if that is not enough."""
        need = 80 * int(n_bytes)
        free, _ = self.ctx.mem_info()
        if free >= need:
            return
        self._out = None
        self._fold(0)
        free, _ = self.ctx.mem_info()
        if free < need:
            raise native.ZotkError(native.ZK_ENOMEM, "a batch of %d bytes needs about %d bytes of device memory, %d are free "
                                   "(table so far: %d entries); use a smaller -m" % (n_bytes, need, free, sum(p[1].n for p in self.parts)))

    def add_device_stream(self, d):
        ctx = self.ctx
        if d.n == 0:
            return
        cap = 2 * d.n
        if self._out is None or self._out[0].n < cap:
            self._out = None
            self._out = (ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32))
        if self.baits is not None:
            # acgt is taken over every read, before the capture filter (kmerize.py:492-493 vs :510-520)
            acgt = ctx.stream_acgt(d, self.K)
            d, _, _ = ctx.capture_filter(d, self.K, self.baits)
            k, c, st = ctx.kmerize(d, self.K, self.flags, self.p, self.seed, out=self._out)
        else:
            k, c, st = ctx.kmerize(d, self.K, self.flags, self.p, self.seed, out=self._out)
            acgt = list(st.acgt)
        for b in range(4):
            self.acgt[b] += acgt[b]
        self.instances += sum(acgt)
        if k.n == 0:
            return
        self.parts.append((0, ctx.copy_of(k), ctx.copy_of(c)))     # exact-size copies: the batch buffers are reused
        while len(self.parts) >= 2 and self.parts[-1][0] == self.parts[-2][0]:
            lv, bk, bc = self.parts.pop()
            _, ak, ac = self.parts.pop()
            nk, nc = ctx.union_sum(ak, ac, bk, bc)
            self.parts.append((lv + 1, nk, nc))

    def _fold(self, keep):
        """union-sum the waiting tables down to one"""
        ctx = self.ctx
        if len(self.parts) > 1:
            k, c, _ = ctx.merge_n([(p[1], p[2]) for p in self.parts])
            top = max(p[0] for p in self.parts) + 1
            self.parts = [(top, ctx.copy_of(k), ctx.copy_of(c))]

    def result(self):
        """(kmers u64[], counts u32[], hist {count: n}) on the host."""
        k, c, h = self.device_result()
        return k.to_host(), c.to_host(), h

    def device_result(self):
        """(kmers, counts) as device arrays + hist, for the device codec."""
        if not self.parts:
            return self.ctx.empty(0, np.uint64).view(0), self.ctx.empty(0, np.uint32).view(0), {}
        self._out = None
        self._fold(0)
        _, k, c = self.parts[0]
        return k, c, self.ctx.hist(c)
