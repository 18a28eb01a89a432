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
    """The process-wide device context (ZOT_DEVICE selects the GPU).  Raises if there is none:
    this build has no CPU path."""
    global _ctx
    if _ctx is None:
        _ctx = native.Context(int(os.environ.get("ZOT_DEVICE", "0")))
    return _ctx


def close():
    global _ctx
    if _ctx is not None:
        _ctx.close()
        _ctx = None


def batch_bytes_for(ctx, requested=None):
    """How much base stream to count per device batch.  zk_kmerize needs about 16 B of sort buffer per
    stream byte, up to 24 B per distinct canonical k-mer for the mirror step and 12 B per output
    entry; 80 B per byte is a safe envelope."""
    free, _ = ctx.mem_info()
    b = max(1 << 20, int(free * 0.8) // 80)
    if requested:
        b = min(b, int(requested))
    return min(b, 6 << 30)


class KmerTable:
    """Sorted distinct k-mers + counts on the device, grown batch by batch."""

    def __init__(self, ctx, K, subsample=None, baits=None):
        self.ctx, self.K = ctx, K
        self.flags = native.KMERIZE_CANONICAL
        self.p, self.seed = 0.0, 0
        if subsample is not None:
            self.flags |= native.KMERIZE_SUBSAMPLE
            self.p, self.seed = subsample
        self.baits = baits            # DeviceArray of sorted both-strand bait k-mers, or None
        self.kmers = None
        self.counts = None
        self.acgt = [0, 0, 0, 0]
        self.instances = 0

    def add_fastq_text(self, text, line_phase):
        """Count one batch of raw FASTQ text (whole lines): parsed on the device (zk_fastq_mask)."""
        d = self.ctx.upload_stream(text)
        if d.n == 0:
            return
        stream, _ = self.ctx.fastq_mask(d, line_phase)
        del d
        self.add_device_stream(stream)

    def add_stream(self, stream_host):
        """Count one batch (uint8 base stream on the host) into the table."""
        self.add_device_stream(self.ctx.upload_stream(stream_host))

    def add_device_stream(self, d):
        ctx = self.ctx
        if d.n == 0:
            return
        if self.baits is not None:
            # acgt is taken over every read, before the capture filter (kmerize.py:492-493 vs :510-520)
            acgt = ctx.stream_acgt(d, self.K)
            d, _, _ = ctx.capture_filter(d, self.K, self.baits)
            k, c, st = ctx.kmerize(d, self.K, self.flags, self.p, self.seed)
        else:
            k, c, st = ctx.kmerize(d, self.K, self.flags, self.p, self.seed)
            acgt = list(st.acgt)
        for b in range(4):
            self.acgt[b] += acgt[b]
        self.instances += sum(acgt)
        if self.kmers is None:
            # keep exact-size copies so the oversized output buffers can go
            self.kmers, self.counts = _compact(ctx, k, c)
        else:
            nk, nc = ctx.union_sum(self.kmers, self.counts, k, c)
            self.kmers, self.counts = _compact(ctx, nk, nc)

    def result(self):
        """(kmers u64[], counts u32[], hist {count: n}) on the host."""
        if self.kmers is None:
            return np.empty(0, np.uint64), np.empty(0, np.uint32), {}
        return self.kmers.to_host(), self.counts.to_host(), self.ctx.hist(self.counts)

    def device_result(self):
        """(kmers, counts) as device arrays + hist, for the device codec."""
        if self.kmers is None:
            return self.ctx.empty(0, np.uint64), self.ctx.empty(0, np.uint32), {}
        return self.kmers, self.counts, self.ctx.hist(self.counts)


def _compact(ctx, k, c):
    """Copy views of oversized buffers into right-sized allocations."""
    lib = ctx.lib
    nk, nc = ctx.empty(k.n, k.dtype), ctx.empty(c.n, c.dtype)
    ctx._check(lib.zk_copy(ctx.h, nk.ptr, k.ptr, k.nbytes))
    ctx._check(lib.zk_copy(ctx.h, nc.ptr, c.ptr, c.nbytes))
    ctx.sync()
    return nk, nc
