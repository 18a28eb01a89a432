"""
Vector members of a k-mer set: numpy arrays <-> codec64 word streams, through the native codec in
libzotk.so (csrc/hostio.cpp).  Reference: zotmer/library/files.py:54-227 -- k-mers are stored as the
codec64 words of their deltas (first delta from 0), counts as the codec64 words of the raw values,
words back to back as little-endian uint64 with no header.
"""
import ctypes as C

import numpy as np

from zotmer_amd.library.timing import Phase

from zotmer_amd import native


class CodecError(ValueError):
    pass


def _enc(values, delta):
    a = np.ascontiguousarray(values, dtype=np.uint64)
    words = np.empty(max(a.size, 1), dtype=np.uint64)
    n = C.c_uint64(0)
    rc = native.load().zk_codec64_encode(a.ctypes.data, a.size, int(delta), words.ctypes.data, words.size, C.byref(n))
    if rc == native.ZK_ERANGE:
        # the reference dies here too (IndexError / struct.error, codec64.py:33-40, files.py:65-83)
        raise CodecError("a value (or k-mer delta) >= 2**60 cannot be stored in the codec64 format")
    if rc != native.ZK_OK:
        raise CodecError("codec64 encode failed (%d)" % rc)
    return words[:n.value]


def _dec(data, delta):
    words = np.frombuffer(data, dtype="<u8")
    lib = native.load()
    n = C.c_uint64(0)
    rc = lib.zk_codec64_count(words.ctypes.data, words.size, C.byref(n))
    if rc != native.ZK_OK:
        raise CodecError("corrupt codec64 stream (unknown tag)")
    out = np.empty(max(n.value, 1), dtype=np.uint64)
    m = C.c_uint64(0)
    rc = lib.zk_codec64_decode(words.ctypes.data, words.size, int(delta), out.ctypes.data, out.size, C.byref(m))
    if rc != native.ZK_OK:
        raise CodecError("codec64 decode failed (%d)" % rc)
    return out[:m.value]


def encode_kmers(kmers):
    """ascending uint64 k-mers -> bytes of the 'kmers' member (files.writeKmers, files.py:143-147)"""
    return _enc(kmers, True).astype("<u8", copy=False).tobytes()


def encode_counts(counts):
    """counts -> bytes of the 'counts' member (files.writeCounts, files.py:155-156)"""
    return _enc(counts, False).astype("<u8", copy=False).tobytes()


def decode_kmers(data):
    """bytes of a 'kmers' member -> uint64 k-mers (files.readKmers, files.py:152-153)"""
    return _dec(data, True)


def decode_counts(data):
    """bytes of a 'counts' member -> uint64 counts (files.readCounts, files.py:158-159)"""
    return _dec(data, False)


def read_kmers(z, name=None):
    """k-mers of a set (files.readKmers, files.py:152-153)"""
    return decode_kmers(z.read((name + "-kmers") if name else "kmers"))


def read_kmers_and_counts(z, name=None):
    """(kmers, counts) of a k-mer set (files.readKmersAndCounts, files.py:219-227)"""
    if name:
        kn, cn = name + "-kmers", name + "-counts"
    else:
        kn, cn = "kmers", "counts"
    k = decode_kmers(z.read(kn))
    c = decode_counts(z.read(cn))
    if len(k) != len(c):
        raise CodecError("k-mer and count vectors differ in length (%d vs %d)" % (len(k), len(c)))
    return k, c


def write_kmers_and_counts(z, kmers, counts, name=None):
    """files.writeKmersAndCounts2 (files.py:209-217): member order kmers, then counts"""
    if name:
        kn, cn = name + "-kmers", name + "-counts"
    else:
        kn, cn = "kmers", "counts"
    z.add(kn, encode_kmers(kmers))
    z.add(cn, encode_counts(counts))


# ---- the same members through the device codec (K11 / K12): values never visit the host ----------------

def _check_codec(fn, *a):
    try:
        return fn(*a)
    except native.ZotkError as e:
        if e.code == native.ZK_ERANGE:
            raise CodecError(str(e))
        raise


def device_encode_kmers(ctx, kmers_dev):
    """ascending uint64 k-mers on the device -> bytes of the 'kmers' member"""
    if kmers_dev.n == 0:
        return b""
    return _check_codec(ctx.codec_encode, kmers_dev, True).to_host().astype("<u8", copy=False).tobytes()


def device_encode_counts(ctx, counts_dev):
    """uint32 / uint64 counts on the device -> bytes of the 'counts' member"""
    if counts_dev.n == 0:
        return b""
    return _check_codec(ctx.codec_encode, counts_dev, False).to_host().astype("<u8", copy=False).tobytes()


def device_write_kmers_and_counts(ctx, z, kmers_dev, counts_dev):
    """files.writeKmersAndCounts2 (files.py:209-217) from device arrays: delta + codec64 on the device (K12), the word streams
    go from device memory to the file without a host copy of our own (Container.add_device)."""
    if kmers_dev.n == 0:
        z.add("kmers", b"")
        z.add("counts", b"")
        return
    with Phase(ctx, "codec64 encode k-mers", kmers_dev.nbytes):
        words = _check_codec(ctx.codec_encode, kmers_dev, True)
    with Phase(ctx, "write k-mers", words.nbytes):
        z.add_device("kmers", ctx, words)
    del words
    with Phase(ctx, "codec64 encode counts", counts_dev.nbytes):
        words = _check_codec(ctx.codec_encode, counts_dev, False)          # 32-bit counts are encoded as they are
    with Phase(ctx, "write counts", words.nbytes):
        z.add_device("counts", ctx, words)


def device_read_kmers(ctx, z):
    """'kmers' member -> uint64 k-mers on the device"""
    if z.member_size("kmers") == 0:
        return ctx.empty(0, np.uint64)
    with Phase(ctx, "read k-mers", z.member_size("kmers")):
        w = z.read_device("kmers", ctx, "<u8")
    with Phase(ctx, "codec64 decode k-mers", w.nbytes):
        return _check_codec(ctx.codec_decode, w, True)


def device_read_kmers_and_counts(ctx, z):
    """(k-mers, counts) of a set as uint64 device arrays"""
    k = device_read_kmers(ctx, z)
    c = _check_codec(ctx.codec_decode, z.read_device("counts", ctx, "<u8"), False) if z.member_size("counts") else ctx.empty(0, np.uint64)
    if k.n != c.n:
        raise CodecError("k-mer and count vectors differ in length (%d vs %d)" % (k.n, c.n))
    return k, c


def device_read_kmers_shard(ctx, z, rank, world, comm):
    """This rank's contiguous piece of a sorted 'kmers' member (position-sharded: rank order = value order) for the
    multi-GPU `zot dist`.  codec64 words decode independently, only the delta transform chains them: every rank decodes
    words [rank, rank + 1) * nw / world, sums its own deltas, the sums are exchanged, and the total of the earlier ranks
    is the base the piece continues from (files.undelta, zotmer/library/files.py:100-110)."""
    nw = z.member_size("kmers") // 8
    w0, w1 = nw * rank // world, nw * (rank + 1) // world
    if w1 > w0:
        words = np.frombuffer(z.read_range("kmers", 8 * w0, 8 * (w1 - w0)), dtype="<u8")
        vals = _check_codec(ctx.codec_decode, ctx.upload(words), False)
        ctx.undelta(vals, 0)
        last = int(vals.view(1, vals.n - 1).to_host()[0]) if vals.n else 0
    else:
        vals, last = ctx.empty(0, np.uint64), 0
    sums = [0] * world
    sums[rank] = last
    sums = comm.all_reduce(sums)
    base = sum(sums[:rank]) & 0xFFFFFFFFFFFFFFFF
    if base and vals.n:
        ctx.add_u64(vals, base)
        ctx.sync()
    return vals
