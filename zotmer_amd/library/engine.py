"""
GPU orchestration shared by the commands: one lazily created zk_ctx, and the batch accumulator that
plays the role of the reference's KmerAccumulator2 + spill files (zotmer/commands/kmerize.py:370-437,
528-553): every batch of reads is counted on the device (zk_kmerize) and union-summed into a resident
sorted table (zk_union_sum); nothing is spilled to disk -- the final arrays do not depend on where
the batches are cut (verified against the reference's -m 1 run, tests/golden).
"""
import os
import sys
import time

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
    if world <= 1 and os.environ.get("ZOT_FORCE_DIST") == "1":
        # rehearse the multi-GPU code path of a command with ONE rank (what a one-GPU box can check on hardware)
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29541")):
            os.environ.setdefault(k, v)
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            from zotmer_amd import parallel
            torch.cuda.set_device(0)
            with parallel.stdout_to_stderr():
                dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
                dist.barrier()
        return dist, 1, 0
    if world <= 1:
        return None, 1, 0
    from zotmer_amd import parallel
    dist = parallel.init_from_env()
    return dist, dist.get_world_size(), dist.get_rank()


def close():
    global _ctx
    for pair in list(_slabs.values()):
        for sl in pair:
            sl.release()
    _slabs.clear()
    if _ctx is not None:
        _ctx.close()
        _ctx = None


def batch_bytes_for(ctx, requested=None, input_bytes=None):
    """How much text / base stream to count per device batch.  zk_kmerize needs about 16 B of sort buffer per stream byte,
    up to 24 B per distinct canonical k-mer for the mirror step, and the table slab holds 12 B per entry; 80 B per byte is
    a safe envelope, and half of the free memory is left for the table that grows while batches are counted.  Without -m
    the batch follows the input size (an eighth of it, between 64 MiB and 2 GiB): device memory costs ~25 ms per GB to
    allocate on this stack, so small inputs should not reserve large workspaces, and large inputs want few merges."""
    free, _ = ctx.mem_info()
    b = max(1 << 20, int(free * 0.5) // 80)
    if requested:
        b = min(b, int(requested))
    elif input_bytes is not None:
        b = min(b, max(64 << 20, min(int(input_bytes) // 8, 2 << 30)))
    return min(b, 6 << 30)


from zotmer_amd.library.timing import Phase as _Phase


class Slab:
    """A pair of device arrays, keys u64[E] and counts u32[E], addressed by the same entry offsets, allocated once per
    context and kept between commands (hipMalloc costs ~25 ms per GB here: table memory must not be reallocated per batch).
    ensure() grows it: with live entries a larger pair is allocated and the prefix copied; with none the old pair is freed
    first, so the peak is the new size only."""

    def __init__(self, ctx):
        self.ctx, self.k, self.c, self.E = ctx, None, None, 0

    def ensure(self, need, live=0, target=None, spare=None):
        """make room for `need` entries, preserving the first `live`; `target` (>= need) is the size to grow to if it fits.
        spare: another slab with nothing live in it (the merge scratch between merges) -- given up first when growing with live
        entries would otherwise not fit (old pair + new pair + a scratch that a slab swap has left at the table's full size)"""
        if need <= self.E:
            return
        ctx = self.ctx
        want = max(int(need), int(target or 0), int(self.E * 1.5) if live else 0)
        free, _ = ctx.mem_info()
        room = free + (0 if live else 12 * self.E)
        if 12 * want + (1 << 30) > room:                      # the comfortable size does not fit: take exactly what is needed
            want = int(need)
        with _Phase(ctx, "table memory -> %d entries" % want):
            if not live:
                self.k = self.c = None
                self.E = 0
            try:
                nk, nc = ctx.empty(want, np.uint64), ctx.empty(want, np.uint32)
            except native.ZotkError:
                # old pair + new pair + a scratch slab that a swap left at the table's full size do not fit together: the scratch slab
                # goes first (it regrows at the next merge -- at 34 ms per GB, so only when there is no other way)
                if not (live and spare is not None and spare.E):
                    raise
                nk = nc = None
                spare.release()
                nk, nc = ctx.empty(want, np.uint64), ctx.empty(want, np.uint32)
            if live:
                ctx._check(ctx.lib.zk_copy(ctx.h, nk.ptr, self.k.ptr, 8 * live))
                ctx._check(ctx.lib.zk_copy(ctx.h, nc.ptr, self.c.ptr, 4 * live))
                ctx.sync()
            self.k, self.c, self.E = nk, nc, want

    def release(self):
        self.k = self.c = None
        self.E = 0


_slabs = {}


def slabs_for(ctx):
    """(table slab, merge scratch) of a context"""
    s = _slabs.get(id(ctx))
    if s is None or s[0].ctx is not ctx:
        s = _slabs[id(ctx)] = (Slab(ctx), Slab(ctx))
    return s


def release_table_memory(ctx):
    for sl in _slabs.pop(id(ctx), ()):
        sl.release()


class KmerTable:
    """Sorted distinct k-mers + counts on the device, grown batch by batch -- the role of the reference's KmerAccumulator2 +
    spill files + mergeNinto (zotmer/commands/kmerize.py:236-304,370-437,528-553).

    Every batch is sorted and counted (zk_kmerize) straight into the top of the table slab; the tables sit on a stack and
    are union-summed pairwise like a binary counter (two tables of the same level make one of the next): the union goes to
    a scratch slab sized for exactly |A u B| (|A n B| is one cheap intersect pass, skipped when the slab holds |A| + |B| anyway)
    and is copied back over its inputs.  So
    n batches cost O(n log n) table traffic, steady state allocates nothing, and peak memory is the live tables plus one
    merge output.  The final arrays do not depend on where the batches are cut (the reference's -m 1 run, tests/golden)."""

    EXACT_FROM = 1 << 25        # merges of at least this many entries size their output by an intersect pass first

    def __init__(self, ctx, K, subsample=None, baits=None):
        self.ctx, self.K = ctx, K
        self.flags = native.KMERIZE_CANONICAL
        self.p, self.seed = 0.0, 0
        if subsample is not None:
            self.flags |= native.KMERIZE_SUBSAMPLE
            self.p, self.seed = subsample
            baits = None              # `if d is not None: ... elif B is not None:` (kmerize.py:494-520): -D wins over -C
        self.baits = baits            # DeviceArray of sorted both-strand bait k-mers, or None
        # Without -D the batches are counted as CANONICAL lists (c = min(x, rc x), one entry per distinct window; half the
        # size of the both-strand table), merged as such, and the strands are rebuilt ONCE at the end (zk_mirror_expand):
        # a batch skips a fifth of its work, the merges move half the bytes.  -D is a predicate on the k-mers of both strands
        # (x and rc x are hashed independently, kmerize.py:495-503), so it keeps the literal both-strand tables.
        self.canonical = subsample is None
        if self.canonical:
            self.flags |= native.KMERIZE_CANONICAL_ONLY
        self.slab, self.scratch = slabs_for(ctx)
        self.stack = []               # [(offset, n, level)] bottom to top, contiguous from offset 0
        self.top = 0
        self.ratio = None             # largest entries-per-stream-byte seen so far (sizes the next batch's output)
        self.expected_bytes = None    # total stream bytes the caller expects to feed (sizes the slab in one step)
        self.seen_bytes = 0
        self.acgt = [0, 0, 0, 0]
        self.instances = 0

    def expect(self, total_stream_bytes):
        """The caller knows how much input is coming: after the first batch the table slab is grown ONCE to the size that
        much input can need at the yield seen so far, instead of geometrically."""
        self.expected_bytes = int(total_stream_bytes)

    def _slab_target(self, need):
        if self.expected_bytes is None or self.ratio is None:
            return None
        rest = max(self.expected_bytes - self.seen_bytes, 0)
        predicted = self.top + int(rest * self.ratio * 1.05) + 65536      # as if later batches shared nothing with earlier ones
        # ... which overshoots on high-coverage data (most k-mers of a later batch are already in the table), and memory
        # costs time to allocate: never more than double in one step
        return max(need, min(predicted, 2 * max(self.slab.E, need)))

    # ---- input forms --------------------------------------------------------------------------------------
    def add_fastq_text(self, text, line_phase):
        """Count one batch of raw FASTQ text (whole lines, host bytes): parsed on the device (zk_fastq_mask)."""
        d = self.ctx.upload_stream(text)
        if d.n == 0:
            return
        stream, _ = self.ctx.fastq_mask(d, line_phase)
        del d
        self.add_device_stream(stream, bound=stream.n)

    def add_stream(self, stream_host):
        """Count one batch (uint8 base stream on the host) into the table."""
        self.add_device_stream(self.ctx.upload_stream(stream_host))

    def add_device_stream(self, d, bound=None):
        """d: base stream on the device.  bound: an upper bound on the entries it can produce (default 2 per byte: one
        window per byte, both strands; FASTQ text gives less than 1 per byte: sequence lines are under half of it)."""
        ctx, slab = self.ctx, self.slab
        if d.n == 0:
            return
        bound = int(bound) if bound else 2 * d.n
        if self.canonical:
            bound = (bound + 1) // 2          # one entry per window, not two
        est = bound if self.ratio is None else min(bound, int(d.n * self.ratio * 1.2) + 65536)
        acgt = None
        n_in = d.n
        if self.baits is not None:
            # acgt is taken over every read, before the capture filter (kmerize.py:492-493 vs :510-520)
            acgt = ctx.stream_acgt(d, self.K)
            d, _, _ = ctx.capture_filter(d, self.K, self.baits)
        while True:
            slab.ensure(self.top + est, self.top, self._slab_target(self.top + est), spare=self.scratch)
            out = (slab.k.view(est, self.top), slab.c.view(est, self.top))
            try:
                with _Phase(ctx, "zk_kmerize"):
                    k, c, st = ctx.kmerize(d, self.K, self.flags, self.p, self.seed, out=out)
                break
            except native.ZotkError as e:
                if e.code != native.ZK_ENOSPC or est >= bound:
                    raise
                est = min(bound, 2 * est)          # the estimate from earlier batches was too small: once more with room
        if acgt is None:
            acgt = list(st.acgt)
        for b in range(4):
            self.acgt[b] += acgt[b]
        self.instances += sum(acgt)
        self.seen_bytes += n_in
        self.ratio = max(self.ratio or 0.0, k.n / float(n_in))
        if k.n == 0:
            return
        self.stack.append((self.top, k.n, 0))
        self.top += k.n
        while len(self.stack) >= 2 and self.stack[-1][2] == self.stack[-2][2]:
            self._merge_top()

    def _merge_top(self):
        """union-sum the two tables on top of the stack into one: into the scratch slab, then back over the inputs (or, at the
        bottom of the stack, the slabs swap)"""
        ctx, slab, scratch = self.ctx, self.slab, self.scratch
        ob, nb, lb = self.stack.pop()
        oa, na, la = self.stack.pop()
        assert oa + na == ob and ob + nb == self.top
        ak, ac = slab.k.view(na, oa), slab.c.view(na, oa)
        bk, bc = slab.k.view(nb, ob), slab.c.view(nb, ob)
        n_out = na + nb
        # ... unless the scratch slab holds |A| + |B| entries as it is: the exact size could only save memory that is already there
        if n_out >= self.EXACT_FROM and scratch.E < n_out:
            with _Phase(ctx, "intersect (size the union)"):
                n_out -= ctx.split(ak, bk)[0]
        # The two tables at the bottom of the stack (the big merges): the union is not copied back -- the two slabs change
        # roles, the scratch slab with the union in it becomes the table slab.  For that it must be as large as the table
        # slab is (or the next batch would have to grow it, with a copy); if memory does not allow that, copy back as above
        # the bottom.
        swap = oa == 0
        # ... as large as the table will have to be, that is: what the input still to come can add at the yield seen so far (when the
        # caller has said how much is coming), never more than the table slab is now -- a slab that an earlier run left larger than
        # this run needs (the strands of the last result were rebuilt in it) is not a size to grow the other one to: 34 ms per GB
        tgt = self._slab_target(n_out)
        want_E = slab.E if tgt is None else min(slab.E, max(tgt, n_out))
        if swap and scratch.E < want_E:
            free, _ = ctx.mem_info()
            if 12 * want_E + (2 << 30) < free + 12 * scratch.E:          # (growing frees the old scratch first: nothing in it is live)
                scratch.ensure(want_E)
            else:
                swap = False
        scratch.ensure(n_out)
        with _Phase(ctx, "union_sum %d + %d" % (na, nb)):
            mk, mc = ctx.union_sum(ak, ac, bk, bc, out=(scratch.k.view(n_out), scratch.c.view(n_out)))
            n = mk.n
            if swap:
                ctx.sync()
                del ak, ac, bk, bc, mk, mc
                slab.k, scratch.k = scratch.k, slab.k
                slab.c, scratch.c = scratch.c, slab.c
                slab.E, scratch.E = scratch.E, slab.E
            else:
                ctx._check(ctx.lib.zk_copy(ctx.h, slab.k.ptr + 8 * oa, mk.ptr, 8 * n))
                ctx._check(ctx.lib.zk_copy(ctx.h, slab.c.ptr + 4 * oa, mc.ptr, 4 * n))
                ctx.sync()
        self.stack.append((oa, n, max(la, lb) + 1))
        self.top = oa + n

    def result(self):
        """(kmers u64[], counts u32[], hist {count: n}) on the host."""
        k, c, h = self.device_result()
        return k.to_host(), c.to_host(), h

    def _folded(self):
        """the one table left after merging whatever waits on the stack: (kmers, counts) views of the table slab"""
        if not self.stack:
            e = self.ctx.empty(0, np.uint64)
            return e.view(0), self.ctx.empty(0, np.uint32).view(0)
        while len(self.stack) > 1:
            self._merge_top()
        off, n, _ = self.stack[0]
        return self.slab.k.view(n, off), self.slab.c.view(n, off)

    def canonical_result(self):
        """(canonical k-mers, counts) before the strands are rebuilt -- what the multi-GPU path exchanges.  Only in
        canonical mode (no -D)."""
        assert self.canonical
        return self._folded()

    def device_result(self):
        """(kmers, counts) of both strands as device arrays (views of the table memory, valid until it is used again) + hist."""
        k, c = self._folded()
        if self.canonical and k.n:
            self.scratch.ensure(2 * k.n)
            with _Phase(self.ctx, "mirror_expand %d" % k.n):
                k, c = self.ctx.mirror_expand(k, c, self.K, out=(self.scratch.k.view(2 * k.n), self.scratch.c.view(2 * k.n)))
        if k.n == 0:
            return k, c, {}
        with _Phase(self.ctx, "hist"):
            h = self.ctx.hist(c)
        return k, c, h


# ---- FASTQ files straight to the device (csrc/ingest.hip) -------------------------------------------------------

def count_fastq_file(ctx, table, path, batch_bytes, take=None):
    """file.readFastq (zotmer/library/file.py:38-52) for a whole file without the host parsing a byte: a zk_source reads the
    file (plain or gzip) ahead of the device into one of two device buffers while the other is being counted; batches are
    cut at line ends on the device, the bytes after the cut are carried to the front of the next buffer; zk_fastq_mask turns
    the text into a base stream and counts the lines.  take(batch index) -> bool selects the batches this process counts
    (multi-GPU: reads are sharded by batch).  Returns the number of records (kmerize.py:527 counts every record)."""
    B = int(batch_bytes)
    lines = 0
    with ctx.source_open(path) as src:
        bufs = [ctx.empty(B + 64, np.uint8), ctx.empty(B + 64, np.uint8)]
        masked = ctx.empty(B + 64, np.uint8)
        cur, carry, index = 0, 0, 0
        src.start(bufs[0], 0, B)
        while True:
            with _Phase(ctx, "wait for the reader"):
                got, eof = src.finish()
            buf, n = bufs[cur], carry + got
            if eof:
                cut = n
            else:
                cut = ctx.last_newline(buf, n)
                if cut == 0:
                    raise IOError("%s: a line longer than the batch size (%d bytes); use a larger -m" % (path, B))
                tail = n - cut
                nxt = bufs[1 - cur]
                if tail:
                    ctx._check(ctx.lib.zk_copy(ctx.h, nxt.ptr, buf.ptr + cut, tail))
                    ctx.sync()
                src.start(nxt, tail, B - tail)          # the next batch streams in while this one is counted
                carry = tail
            if eof and cut and buf.view(1, cut - 1).to_host()[0] != 10:
                ctx._check(ctx.lib.zk_upload(ctx.h, buf.ptr + cut, b"\n", 1))      # the last line counts without a terminator
                cut += 1
            if cut:
                with _Phase(ctx, "fastq_mask", cut):
                    stream, nl = ctx.fastq_mask(buf.view(cut), lines % 4, out=masked.view(cut))
                if eof and (lines + nl) % 4:
                    # an incomplete final record: file.readFastq drops it (file.py:51-52) -- find where its lines start and
                    # mask again without them (malformed input only; the tail is scanned on the host)
                    extra = (lines + nl) % 4
                    m = min(cut, 1 << 22)
                    t = buf.view(m, cut - m).to_host().tobytes()
                    pos = len(t)
                    for _ in range(extra):
                        pos = t.rfind(b"\n", 0, pos - 1) + 1 if pos > 0 else 0
                    if pos == 0 and m < cut:
                        raise IOError("%s: the incomplete final record is longer than 4 MiB" % path)
                    cut = cut - m + pos
                    nl -= extra
                    if cut:
                        stream, _ = ctx.fastq_mask(buf.view(cut), lines % 4, out=masked.view(cut))
                if cut and (take is None or take(index)):
                    table.add_device_stream(stream, bound=cut)
                lines += nl
                index += 1
            if eof:
                break
            cur = 1 - cur
    return lines // 4


def measure_prep(ctx, kmers, shift):
    """Measure.prep of `zot dist` (commands/dist.py:43-49): xs = uniq(x >> shift).  At shift 0 -- the file's own K, the usual case --
    that is the identity on a valid set, whose k-mers ascend strictly (library/files.py:54-110): one read-only pass checks
    exactly that (zk_first_descent: 8 B/k-mer read, nothing written) and the decoded array is used as it is; only a set that
    breaks the format's invariant, or a projection to a shorter K, goes through the copying pass (zk_project_dedupe)."""
    if shift == 0 and ctx.first_descent(kmers) == kmers.n:
        return kmers
    return ctx.project_dedupe(kmers, shift)

