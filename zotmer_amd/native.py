"""
ctypes binding of libzotk.so (include/zotk.h) -- the only way Python reaches the HIP kernels.

There is no CPU fallback: if the shared library is missing, or no MI355X is visible, the calls
below raise.  The library is built in-tree by `__graft_entry__.build()` (or
`make -C zotmer_amd/csrc`) so that it travels with the source tree.

Device memory is handled through small `DeviceArray` objects (pointer + dtype + length) that
free themselves; numpy arrays go up with `Context.upload` and come back with
`DeviceArray.to_host`.  torch tensors can be used instead: pass `tensor.data_ptr()` wrapped in
`DeviceArray.borrow(...)`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZOTK_LIB: load a differently built libzotk.so (diagnostic builds, tools/stamps.py); same ABI required
LIB_PATH = os.environ.get("ZOTK_LIB") or os.path.join(_HERE, "libzotk.so")

ZK_OK, ZK_EINVAL, ZK_ENOMEM, ZK_EHIP, ZK_ENOSPC, ZK_EOVERFLOW, ZK_EINTERNAL, ZK_ERANGE = 0, -1, -2, -3, -4, -5, -6, -7
KMERIZE_CANONICAL, KMERIZE_BOTH, KMERIZE_SUBSAMPLE, KMERIZE_CANONICAL_ONLY = 0, 1, 2, 4
DEFAULT_TAG_WORDS = 2          # zk_tune(ZK_TUNE_TAG_WORDS) as the library starts (csrc/internal.hpp)

_ERRNAMES = {-1: "ZK_EINVAL", -2: "ZK_ENOMEM", -3: "ZK_EHIP", -4: "ZK_ENOSPC", -5: "ZK_EOVERFLOW",
             -6: "ZK_EINTERNAL", -7: "ZK_ERANGE"}


class ZotkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (_ERRNAMES.get(code, code), msg))
        self.code = code


class KmerizeStats(C.Structure):
    _fields_ = [("n_windows", C.c_uint64), ("n_instances", C.c_uint64), ("n_unique", C.c_uint64),
                ("n_canonical", C.c_uint64), ("acgt", C.c_uint64 * 4)]


# name -> (restype, argtypes): every symbol include/zotk.h declares
_vp, _u64, _u32, _i, _d = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double
_pu64 = C.POINTER(C.c_uint64)
SIGNATURES = {
    "zk_create": (_vp, [_i, _u64]),
    "zk_destroy": (None, [_vp]),
    "zk_last_error": (C.c_char_p, [_vp]),
    "zk_set_stream": (_i, [_vp, _vp]),
    "zk_get_stream": (_vp, [_vp]),
    "zk_sync": (_i, [_vp]),
    "zk_reserve": (_i, [_vp, _u64]),
    "zk_release_workspace": (_i, [_vp]),
    "zk_mem_info": (_i, [_vp, _pu64, _pu64]),
    "zk_alloc": (_i, [_vp, _u64, C.POINTER(_vp)]),
    "zk_free": (_i, [_vp, _vp]),
    "zk_upload": (_i, [_vp, _vp, _vp, _u64]),
    "zk_download": (_i, [_vp, _vp, _vp, _u64]),
    "zk_copy": (_i, [_vp, _vp, _vp, _u64]),
    "zk_host_alloc": (_i, [_vp, _u64, C.POINTER(_vp)]),
    "zk_host_free": (_i, [_vp, _vp]),
    "zk_upload_async": (_i, [_vp, _vp, _vp, _u64]),
    "zk_tune": (_i, [_vp, _i, _i]),
    "zk_debug_buffer": (_i, [_vp, _vp]),
    "zk_profile": (_i, [_vp, _i]),
    "zk_profile_read": (_i, [_vp, _i, _pu64, C.POINTER(C.c_double), _pu64]),
    "zk_pack_reads": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "zk_encode": (_i, [_vp, _vp, _u64, _i, _i, _vp, _u64, _pu64, _pu64]),
    "zk_capture_filter": (_i, [_vp, _vp, _u64, _i, _vp, _u64, _vp, _pu64, _pu64]),
    "zk_subsample": (_i, [_vp, _vp, _u64, _u64, _d, _vp, _u64, _pu64]),
    "zk_can": (_i, [_vp, _i, _vp, _u64, _vp]),
    "zk_sort_keys": (_i, [_vp, _vp, _u64, _i]),
    "zk_sort_pairs": (_i, [_vp, _vp, _vp, _u64, _i]),
    "zk_rle": (_i, [_vp, _vp, _u64, _vp, _vp, _u64, _pu64]),
    "zk_sort_count": (_i, [_vp, _vp, _u64, _i, _vp, _vp, _u64, _pu64]),
    "zk_kmerize": (_i, [_vp, _vp, _u64, _i, _i, _d, _u64, _vp, _vp, _u64, C.POINTER(KmerizeStats)]),
    "zk_mirror_expand": (_i, [_vp, _vp, _vp, _u64, _i, _vp, _vp, _u64, _pu64]),
    "zk_hist": (_i, [_vp, _vp, _i, _u64, _pu64, _pu64, _u64, _pu64]),
    "zk_widen_counts": (_i, [_vp, _vp, _vp, _u64]),
    "zk_union_sum": (_i, [_vp, _vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _i, _u64, _pu64, _pu64]),
    "zk_merge_n": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), _pu64, _vp, _vp, _i, _u64, _pu64, _pu64]),
    "zk_project_dedupe": (_i, [_vp, _vp, _u64, _i, _vp, _u64, _pu64]),
    "zk_project": (_i, [_vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _u64, _pu64]),
    "zk_sample": (_i, [_vp, _vp, _vp, _u64, _u64, _d, _vp, _vp, _u64, _pu64]),
    "zk_split": (_i, [_vp, _vp, _u64, _vp, _u64, _pu64]),
    "zk_lower_bound": (_i, [_vp, _vp, _u64, _pu64, _u32, _pu64]),
    "zk_trim": (_i, [_vp, _vp, _vp, _i, _u64, _u64, _u64, _vp, _vp, _u64, _pu64]),
    "zk_codec64_encode": (_i, [_vp, _u64, _i, _vp, _u64, _pu64]),
    "zk_codec64_count": (_i, [_vp, _u64, _pu64]),
    "zk_codec64_decode": (_i, [_vp, _u64, _i, _vp, _u64, _pu64]),
    "zk_codec64_encode_dev": (_i, [_vp, _vp, _u64, _i, _vp, _u64, _pu64]),
    "zk_codec64_encode_u32_dev": (_i, [_vp, _vp, _u64, _vp, _u64, _pu64]),
    "zk_codec64_decode_dev": (_i, [_vp, _vp, _u64, _i, _vp, _u64, _pu64]),
    "zk_fastq_mask": (_i, [_vp, _vp, _u64, _u32, _vp, _pu64]),
    "zk_source_open": (_vp, [_vp, C.c_char_p, _i]),
    "zk_source_is_gzip": (_i, [_vp]),
    "zk_source_start": (_i, [_vp, _vp, _u64]),
    "zk_source_finish": (_i, [_vp, _pu64, C.POINTER(_i)]),
    "zk_source_close": (None, [_vp]),
    "zk_last_newline": (_i, [_vp, _vp, _u64, _pu64]),
    "zk_device_to_file": (_i, [_vp, _vp, _u64, _i, _u64, _i]),
    "zk_file_to_device": (_i, [_vp, _i, _u64, _u64, _vp, _i]),
    "zk_undelta": (_i, [_vp, _vp, _u64, _u64]),
    "zk_add_u64": (_i, [_vp, _vp, _u64, _u64]),
    "zk_parse_fastq": (_i, [_vp, _u64, _i, _pu64, _vp, _u64, _pu64, _pu64]),
    "zk_parse_fasta": (_i, [_vp, _u64, _i, _pu64, _vp, _u64, _pu64, _pu64]),
    "zk_synth_reads": (_i, [_vp, _u64, _u64, _u64, _i, _u64, _u32, _u32, _vp]),
    "zk_checksum": (_i, [_vp, _vp, _vp, _u64, _pu64]),
    "zk_checksum_counts": (_i, [_vp, _vp, _vp, _i, _u64, _pu64]),
    "zk_first_descent": (_i, [_vp, _vp, _u64, _pu64]),
    "zk_synth_keys": (_i, [_vp, _u64, _u64, _u64, _i, _u64, _u64, _u64, _vp]),
    "zk_synth_counts": (_i, [_vp, _u64, _vp, _u64, _vp]),
    "zk_hash_partition": (_i, [_vp, _vp, _vp, _i, _u64, _i, _u64, _vp, _vp, _pu64]),
    "zk_comm_unique_id": (_i, [_vp]),
    "zk_comm_init": (_i, [_vp, _i, _i, _vp]),
    "zk_comm_destroy": (_i, [_vp]),
    "zk_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "zk_all_to_all_v": (_i, [_vp, _vp, _pu64, _pu64, _vp, _pu64, _pu64, _i]),
    "zk_allreduce_u64": (_i, [_vp, _pu64, _u64, _i]),
    "zk_comm_plan": (_i, [_i, _i, _pu64, _pu64, _pu64, _pu64, _i, _u64, _i, _vp, _u64, _pu64]),
    "zk_stream_checksum": (_i, [_vp, _vp, _u64, _i, _pu64]),
}

_lib = None


def load():
    """dlopen libzotk.so and bind every declared symbol.  Raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZotkError(ZK_EINTERNAL, "%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                          "or `make -C zotmer_amd/csrc` (needs hipcc; there is no CPU fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class DeviceArray:
    """A typed view of device memory: .ptr, .dtype, .n (elements).  Owns the allocation unless borrowed."""

    def __init__(self, ctx, ptr, dtype, n, owned=True, keep=None):
        self.ctx, self.ptr, self.dtype, self.n, self.owned, self._keep = ctx, ptr, np.dtype(dtype), int(n), owned, keep

    @classmethod
    def borrow(cls, ctx, ptr, dtype, n, keep=None):
        return cls(ctx, ptr, dtype, n, owned=False, keep=keep)

    @property
    def nbytes(self):
        return self.n * self.dtype.itemsize

    def view(self, n, offset=0):
        """First n elements (from element `offset`) as a borrowed array that keeps this one alive."""
        return DeviceArray(self.ctx, self.ptr + offset * self.dtype.itemsize, self.dtype, n, owned=False, keep=self)

    def to_host(self, n=None):
        n = self.n if n is None else int(n)
        out = np.empty(n, dtype=self.dtype)
        if n:
            self.ctx._check(load().zk_download(self.ctx.h, out.ctypes.data, self.ptr, n * self.dtype.itemsize))
        return out

    def free(self):
        if self.owned and self.ptr and self.ctx.h:
            load().zk_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Source:
    """zk_source: a file (plain or gzip) read ahead of the device through page-locked buffers and a copy stream."""

    def __init__(self, ctx, path, threads):
        self.ctx = ctx
        self.h = ctx.lib.zk_source_open(ctx.h, os.fsencode(path), int(threads))
        if not self.h:
            raise IOError(ctx.lib.zk_last_error(ctx.h).decode(errors="replace"))
        self.gzip = bool(ctx.lib.zk_source_is_gzip(self.h))

    def start(self, dst, offset, cap):
        """begin reading up to cap bytes into dst[offset:] (a uint8 DeviceArray); returns at once"""
        self.ctx._check(self.ctx.lib.zk_source_start(self.h, dst.ptr + int(offset), int(cap)))

    def finish(self):
        """wait for the request -> (bytes that arrived, end of input reached)"""
        n, eof = C.c_uint64(0), C.c_int(0)
        self.ctx._check(self.ctx.lib.zk_source_finish(self.h, C.byref(n), C.byref(eof)))
        return n.value, bool(eof.value)

    def close(self):
        if self.h:
            self.ctx.lib.zk_source_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Context:
    """One zk_ctx: one GPU, one stream."""

    def __init__(self, device=0, workspace_bytes=0):
        self.lib = load()
        self.h = self.lib.zk_create(device, workspace_bytes)
        if not self.h:
            why = self.lib.zk_last_error(None).decode(errors="replace")
            raise ZotkError(ZK_EHIP, "zk_create(%d) failed: %s -- no usable MI355X (there is no CPU fallback)" % (device, why))
        self.device = device

    def close(self):
        if self.h:
            self.lib.zk_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != ZK_OK:
            raise ZotkError(rc, self.lib.zk_last_error(self.h).decode(errors="replace"))

    # ---- memory ---------------------------------------------------------------------------
    def mem_info(self):
        f, t = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.zk_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def release_workspace(self):
        self._check(self.lib.zk_release_workspace(self.h))

    def reserve(self, nbytes):
        self._check(self.lib.zk_reserve(self.h, int(nbytes)))

    def sync(self):
        self._check(self.lib.zk_sync(self.h))

    def empty(self, n, dtype):
        dt = np.dtype(dtype)
        p = C.c_void_p(0)
        self._check(self.lib.zk_alloc(self.h, max(int(n), 1) * dt.itemsize, C.byref(p)))
        return DeviceArray(self, p.value, dt, n)

    def upload(self, arr, dtype=None):
        a = np.ascontiguousarray(arr, dtype=dtype)
        d = self.empty(a.size, a.dtype)
        if a.size:
            self._check(self.lib.zk_upload(self.h, d.ptr, a.ctypes.data, a.nbytes))
        return d

    def pinned(self, nbytes):
        """A page-locked uint8 host buffer as a numpy array (freed with the returned object)."""
        p = C.c_void_p(0)
        self._check(self.lib.zk_host_alloc(self.h, int(nbytes), C.byref(p)))
        ctx = self

        class _Pinned:
            def __init__(s):
                s.ptr, s.n = p.value, int(nbytes)
                s.array = np.ctypeslib.as_array((C.c_uint8 * max(s.n, 1)).from_address(s.ptr))[:s.n]

            def free(s):
                if s.ptr and ctx.h:
                    s.array = None
                    ctx.lib.zk_host_free(ctx.h, s.ptr)
                s.ptr = None

            def __del__(s):
                try:
                    s.free()
                except Exception:
                    pass
        return _Pinned()

    def upload_async(self, dst, src_ptr, nbytes):
        self._check(self.lib.zk_upload_async(self.h, dst.ptr, src_ptr, int(nbytes)))

    # ---- ingest: files <-> device memory (csrc/ingest.hip) -------------------------------------------------
    IO_THREADS = int(os.environ.get("ZOT_IO_THREADS", 0)) or max(1, min(8, (os.cpu_count() or 2) - 1))

    def source_open(self, path, threads=None):
        return Source(self, path, threads or self.IO_THREADS)

    def last_newline(self, text, n):
        """position just after the last newline of the first n bytes of a device text buffer (0: none)"""
        cut = C.c_uint64(0)
        self._check(self.lib.zk_last_newline(self.h, text.ptr, int(n), C.byref(cut)))
        return cut.value

    def device_to_file(self, arr, fileno, offset, threads=None):
        self._check(self.lib.zk_device_to_file(self.h, arr.ptr, arr.nbytes, int(fileno), int(offset), threads or self.IO_THREADS))

    def file_to_device(self, fileno, offset, nbytes, dtype=np.uint8, threads=None):
        dt = np.dtype(dtype)
        out = self.empty(nbytes // dt.itemsize, dt)
        self._check(self.lib.zk_file_to_device(self.h, int(fileno), int(offset), int(nbytes), out.ptr, threads or self.IO_THREADS))
        return out

    def upload_stream(self, data):
        """bytes / uint8 array -> device base stream (zk_alloc memory is 256-byte aligned)."""
        a = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else np.asarray(data, np.uint8)
        return self.upload(a)

    # ---- per-launch timing (HIP events on the ctx stream) -----------------------------------
    PROF_TAGS = {"hist_stream": 1, "hist_array": 2, "pass_stream": 3, "pass_keys": 4, "pass_pairs": 5, "rle": 6,
                 "union_sum": 7, "select": 8, "mirror": 9, "intersect": 10, "count_hist": 11, "pass_packed": 12, "sample": 13, "tile_sort": 14}

    def tune(self, sort_variant=None, pairs_variant=None, short_sort=None, side_div=None, xcd_group=None, comm_chunk=None,
             early_collapse=None, packed_pairs=None, wide_tiles=None, stream_pass=None, stream_ranges=None, tag_words=None,
             dedupe_variant=None, dedupe_limit=None, dedupe_bits=None, comm_self_loop=None, tag_pass=None, kway=None, tile_sort=None):
        if tile_sort is not None:
            self._check(self.lib.zk_tune(self.h, 19, int(tile_sort)))
        if kway is not None:
            self._check(self.lib.zk_tune(self.h, 18, int(kway)))
        if tag_pass is not None:
            self._check(self.lib.zk_tune(self.h, 17, int(tag_pass)))
        if comm_self_loop is not None:
            self._check(self.lib.zk_tune(self.h, 16, int(comm_self_loop)))
        if dedupe_bits is not None:
            self._check(self.lib.zk_tune(self.h, 15, int(dedupe_bits)))
        if dedupe_variant is not None:
            self._check(self.lib.zk_tune(self.h, 13, int(dedupe_variant)))
        if dedupe_limit is not None:
            self._check(self.lib.zk_tune(self.h, 14, int(dedupe_limit)))
        if tag_words is not None:
            self._check(self.lib.zk_tune(self.h, 12, int(tag_words)))
        if stream_pass is not None:
            self._check(self.lib.zk_tune(self.h, 10, int(stream_pass)))
        if stream_ranges is not None:
            self._check(self.lib.zk_tune(self.h, 11, int(stream_ranges)))
        if wide_tiles is not None:
            self._check(self.lib.zk_tune(self.h, 9, int(wide_tiles)))
        if packed_pairs is not None:
            self._check(self.lib.zk_tune(self.h, 8, int(packed_pairs)))
        if early_collapse is not None:
            self._check(self.lib.zk_tune(self.h, 7, int(early_collapse)))
        if comm_chunk is not None:
            self._check(self.lib.zk_tune(self.h, 6, int(comm_chunk)))
        if xcd_group is not None:
            self._check(self.lib.zk_tune(self.h, 5, int(xcd_group)))
        if short_sort is not None:
            self._check(self.lib.zk_tune(self.h, 3, int(short_sort)))
        if side_div is not None:
            self._check(self.lib.zk_tune(self.h, 4, int(side_div)))
        if sort_variant is not None:
            self._check(self.lib.zk_tune(self.h, 1, int(sort_variant)))
        if pairs_variant is not None:
            self._check(self.lib.zk_tune(self.h, 2, int(pairs_variant)))

    def profile(self, enable=True):
        self._check(self.lib.zk_profile(self.h, int(enable)))

    def profile_read(self):
        """{kernel: dict(launches, ms, bytes)} for everything recorded since profile(True)."""
        out = {}
        for name, tag in self.PROF_TAGS.items():
            n, b, ms = C.c_uint64(0), C.c_uint64(0), C.c_double(0)
            self._check(self.lib.zk_profile_read(self.h, tag, C.byref(n), C.byref(ms), C.byref(b)))
            if n.value:
                out[name] = dict(launches=n.value, ms=ms.value, bytes=b.value)
        return out

    # ---- kernels ------------------------------------------------------------------------------
    def pack_reads(self, bases, offs):
        n_reads = offs.n - 1
        total = int(offs.to_host()[-1]) if offs.n else 0
        out = self.empty(total + n_reads, np.uint8)
        self._check(self.lib.zk_pack_reads(self.h, bases.ptr, offs.ptr, n_reads, out.ptr))
        return out

    def encode(self, stream, K, both=True):
        cap = stream.n * (2 if both else 1)
        out = self.empty(cap, np.uint64)
        n = C.c_uint64(0)
        acgt = (C.c_uint64 * 4)()
        self._check(self.lib.zk_encode(self.h, stream.ptr, stream.n, K, int(both), out.ptr, cap, C.byref(n), acgt))
        return out.view(n.value), [int(v) for v in acgt]

    def subsample(self, kmers, seed, p):
        out = self.empty(kmers.n, np.uint64)
        n = C.c_uint64(0)
        self._check(self.lib.zk_subsample(self.h, kmers.ptr, kmers.n, int(seed), float(p), out.ptr, kmers.n, C.byref(n)))
        return out.view(n.value)

    def can(self, K, kmers):
        """basics.can per element: the strand with the smaller murmer(., 17)"""
        out = self.empty(kmers.n, np.uint64)
        self._check(self.lib.zk_can(self.h, int(K), kmers.ptr, kmers.n, out.ptr))
        self.sync()
        return out

    def sort_keys(self, keys, key_bits):
        self._check(self.lib.zk_sort_keys(self.h, keys.ptr, keys.n, key_bits))
        return keys

    def sort_pairs(self, keys, vals, key_bits):
        self._check(self.lib.zk_sort_pairs(self.h, keys.ptr, vals.ptr, keys.n, key_bits))
        return keys, vals

    def rle(self, sorted_keys, in_place=False):
        uniq = sorted_keys if in_place else self.empty(sorted_keys.n, np.uint64)
        cnt = self.empty(sorted_keys.n, np.uint32)
        n = C.c_uint64(0)
        self._check(self.lib.zk_rle(self.h, sorted_keys.ptr, sorted_keys.n, uniq.ptr, cnt.ptr, sorted_keys.n, C.byref(n)))
        return uniq.view(n.value), cnt.view(n.value)

    def sort_count(self, keys, key_bits):
        uniq = self.empty(keys.n, np.uint64)
        cnt = self.empty(keys.n, np.uint32)
        n = C.c_uint64(0)
        self._check(self.lib.zk_sort_count(self.h, keys.ptr, keys.n, key_bits, uniq.ptr, cnt.ptr, keys.n, C.byref(n)))
        return uniq.view(n.value), cnt.view(n.value)

    def kmerize(self, stream, K, flags=KMERIZE_CANONICAL, p=0.0, seed=0, cap=None, out=None):
        """-> (kmers DeviceArray u64, counts DeviceArray u32, KmerizeStats).  `out` = (kmers, counts)
        preallocated arrays to write into (their length is the capacity)."""
        if out is None:
            cap = int(cap) if cap is not None else 2 * stream.n
            ok, oc = self.empty(cap, np.uint64), self.empty(cap, np.uint32)
        else:
            ok, oc = out
            cap = min(ok.n, oc.n)
        st = KmerizeStats()
        self._check(self.lib.zk_kmerize(self.h, stream.ptr, stream.n, K, flags, float(p), int(seed), ok.ptr, oc.ptr, cap, C.byref(st)))
        return ok.view(st.n_unique), oc.view(st.n_unique), st

    def mirror_expand(self, ck, cc, K, out=None):
        """counted canonical list (zk_kmerize with KMERIZE_CANONICAL_ONLY) -> (kmers, counts) of both strands"""
        if out is None:
            ok, oc = self.empty(2 * ck.n, np.uint64), self.empty(2 * ck.n, np.uint32)
        else:
            ok, oc = out
        n = C.c_uint64(0)
        self._check(self.lib.zk_mirror_expand(self.h, ck.ptr, cc.ptr, ck.n, int(K), ok.ptr, oc.ptr, min(ok.n, oc.n), C.byref(n)))
        return ok.view(n.value), oc.view(n.value)

    def hist(self, counts):
        bits = counts.dtype.itemsize * 8
        cap = 1 << 16
        while True:
            vals = np.empty(cap, dtype=np.uint64)
            freq = np.empty(cap, dtype=np.uint64)
            n = C.c_uint64(0)
            rc = self.lib.zk_hist(self.h, counts.ptr, bits, counts.n, vals.ctypes.data_as(_pu64), freq.ctypes.data_as(_pu64), cap, C.byref(n))
            if rc == ZK_ENOSPC and cap < (1 << 26):
                cap *= 8
                continue
            self._check(rc)
            return {int(v): int(f) for v, f in zip(vals[:n.value], freq[:n.value])}

    def widen(self, counts32):
        out = self.empty(counts32.n, np.uint64)
        self._check(self.lib.zk_widen_counts(self.h, counts32.ptr, out.ptr, counts32.n))
        return out

    def union_sum(self, xk, xc, yk, yc, want_acgt=False, out=None):
        bits = xc.dtype.itemsize * 8
        assert yc.dtype == xc.dtype
        if out is None:
            cap = xk.n + yk.n
            ok, oc = self.empty(cap, np.uint64), self.empty(cap, xc.dtype)
        else:
            ok, oc = out
            cap = min(ok.n, oc.n)
        n = C.c_uint64(0)
        acgt = (C.c_uint64 * 4)()
        self._check(self.lib.zk_union_sum(self.h, xk.ptr, xc.ptr, xk.n, yk.ptr, yc.ptr, yk.n, ok.ptr, oc.ptr, bits, cap,
                                          C.byref(n), acgt if want_acgt else None))
        r = (ok.view(n.value), oc.view(n.value))
        return r + ([int(v) for v in acgt],) if want_acgt else r

    def merge_n(self, sets, out=None):
        """sets = [(kmers u64 DeviceArray, counts u32|u64 DeviceArray), ...] -> (kmers, counts, acgt_weighted).
        out = (kmers, counts) preallocated arrays to write into (their length is the capacity)."""
        k = len(sets)
        cdt = sets[0][1].dtype if k else np.dtype(np.uint64)
        assert all(s[1].dtype == cdt for s in sets)
        pk = (_vp * k)(*[s[0].ptr for s in sets])
        pc = (_vp * k)(*[s[1].ptr for s in sets])
        ns = (C.c_uint64 * k)(*[s[0].n for s in sets])
        if out is None:
            cap = sum(s[0].n for s in sets)
            ok, oc = self.empty(cap, np.uint64), self.empty(cap, cdt)
        else:
            ok, oc = out
            cap = min(ok.n, oc.n)
        n = C.c_uint64(0)
        acgt = (C.c_uint64 * 4)()
        self._check(self.lib.zk_merge_n(self.h, k, pk, pc, ns, ok.ptr, oc.ptr, cdt.itemsize * 8, cap, C.byref(n), acgt))
        return ok.view(n.value), oc.view(n.value), [int(v) for v in acgt]

    def project_dedupe(self, kmers, shift):
        out = self.empty(kmers.n, np.uint64)
        n = C.c_uint64(0)
        self._check(self.lib.zk_project_dedupe(self.h, kmers.ptr, kmers.n, shift, out.ptr, kmers.n, C.byref(n)))
        return out.view(n.value)

    def split(self, x, y):
        abc = (C.c_uint64 * 3)()
        self._check(self.lib.zk_split(self.h, x.ptr, x.n, y.ptr, y.n, abc))
        return tuple(int(v) for v in abc)

    def codec_encode(self, values, delta):
        """uint64 (or, delta=False, uint32) device values -> device codec64 words (delta=True: ascending k-mers, stored as
        differences).  The word buffer starts at half a word per value (device memory costs ~25 ms per GB to allocate: sorted
        k-mers pack three to a word, counts six) and is grown to the worst case, one word per value, only if that is too small."""
        n = C.c_uint64(0)
        for cap in ((values.n + 1) // 2 + 1024, values.n):
            cap = min(cap, values.n)
            words = self.empty(cap, np.uint64)
            if values.dtype.itemsize == 4:
                if delta:
                    raise ValueError("delta coding is for 64-bit k-mers")
                rc = self.lib.zk_codec64_encode_u32_dev(self.h, values.ptr, values.n, words.ptr, words.n, C.byref(n))
            else:
                rc = self.lib.zk_codec64_encode_dev(self.h, values.ptr, values.n, int(delta), words.ptr, words.n, C.byref(n))
            if rc == ZK_ENOSPC and cap < values.n:
                del words
                continue
            self._check(rc)
            return words.view(n.value)

    def codec_decode(self, words, delta, n_values=None):
        """device codec64 words -> uint64 device values (delta=True also undoes the k-mer differences)."""
        n = C.c_uint64(0)
        if n_values is None:                 # a first pass with no output only counts
            rc = self.lib.zk_codec64_decode_dev(self.h, words.ptr, words.n, 0, None, 0, C.byref(n))
            if rc not in (ZK_OK, ZK_ENOSPC):
                self._check(rc)
            n_values = n.value
        out = self.empty(n_values, np.uint64)
        self._check(self.lib.zk_codec64_decode_dev(self.h, words.ptr, words.n, int(delta), out.ptr, int(n_values), C.byref(n)))
        return out.view(n.value)

    def undelta(self, vals, base=0):
        """in-place prefix sum of decoded k-mer deltas, continued from `base` (files.undelta)"""
        self._check(self.lib.zk_undelta(self.h, vals.ptr, vals.n, int(base) & 0xFFFFFFFFFFFFFFFF))
        return vals

    def add_u64(self, vals, x):
        self._check(self.lib.zk_add_u64(self.h, vals.ptr, vals.n, int(x) & 0xFFFFFFFFFFFFFFFF))
        return vals

    def fastq_mask(self, text, line_phase=0, out=None):
        """FASTQ text on the device -> (base stream of the same length, number of newlines)."""
        out = out if out is not None else self.empty(text.n, np.uint8)
        n = C.c_uint64(0)
        self._check(self.lib.zk_fastq_mask(self.h, text.ptr, text.n, int(line_phase) & 3, out.ptr, C.byref(n)))
        return out, n.value

    def lower_bound(self, sorted_keys, queries):
        q = np.ascontiguousarray(queries, dtype=np.uint64)
        pos = np.zeros(len(q), dtype=np.uint64)
        self._check(self.lib.zk_lower_bound(self.h, sorted_keys.ptr, sorted_keys.n, q.ctypes.data_as(_pu64), len(q),
                                            pos.ctypes.data_as(_pu64)))
        return pos if isinstance(queries, np.ndarray) else [int(p) for p in pos]

    def project(self, ref, kmers, counts):
        ok, oc = self.empty(kmers.n, np.uint64), self.empty(kmers.n, np.uint64)
        n = C.c_uint64(0)
        self._check(self.lib.zk_project(self.h, ref.ptr, ref.n, kmers.ptr, counts.ptr, kmers.n, ok.ptr, oc.ptr, kmers.n, C.byref(n)))
        return ok.view(n.value), oc.view(n.value)

    def sample(self, kmers, counts, seed, p):
        ok, oc = self.empty(kmers.n, np.uint64), self.empty(kmers.n, np.uint64)
        n = C.c_uint64(0)
        self._check(self.lib.zk_sample(self.h, kmers.ptr, counts.ptr, kmers.n, int(seed), float(p), ok.ptr, oc.ptr, kmers.n, C.byref(n)))
        return ok.view(n.value), oc.view(n.value)

    def trim(self, kmers, counts, lo, hi=0):
        bits = counts.dtype.itemsize * 8
        ok, oc = self.empty(kmers.n, np.uint64), self.empty(kmers.n, counts.dtype)
        n = C.c_uint64(0)
        self._check(self.lib.zk_trim(self.h, kmers.ptr, counts.ptr, bits, kmers.n, int(lo), int(hi), ok.ptr, oc.ptr, kmers.n, C.byref(n)))
        return ok.view(n.value), oc.view(n.value)

    def synth_reads(self, seed, first, count, L, genome=0, sub_thr=0, n_thr=0, out=None):
        out = out if out is not None else self.empty(count * (L + 1), np.uint8)
        self._check(self.lib.zk_synth_reads(self.h, int(seed), int(first), int(count), int(L), int(genome), int(sub_thr), int(n_thr), out.ptr))
        return out

    def checksum(self, kmers, counts=None):
        s = (C.c_uint64 * 3)()
        self._check(self.lib.zk_checksum(self.h, kmers.ptr, counts.ptr if counts is not None else None, kmers.n, s))
        return tuple(int(v) for v in s)

    def first_descent(self, kmers):
        """first index whose k-mer is not above its predecessor; kmers.n if strictly ascending"""
        r = C.c_uint64(0)
        self._check(self.lib.zk_first_descent(self.h, kmers.ptr, kmers.n, C.byref(r)))
        return int(r.value)

    def checksum_counts(self, kmers, counts):
        """zk_checksum over 32- or 64-bit counts"""
        s = (C.c_uint64 * 3)()
        self._check(self.lib.zk_checksum_counts(self.h, kmers.ptr, counts.ptr if counts is not None else None,
                                                counts.dtype.itemsize * 8 if counts is not None else 64, kmers.n, s))
        return tuple(int(v) for v in s)

    def synth_set(self, seed, first, count, key_bits, mul=1, add=0, mod=1 << 62, counts=True):
        """A synthetic sorted k-mer set (zotmer_amd/synth.py set_keys / set_counts): sorted distinct keys of the
        affine pool walk, and geometric 64-bit counts.  -> (keys u64, counts u64 | None)"""
        raw = self.empty(count, np.uint64)
        self._check(self.lib.zk_synth_keys(self.h, int(seed), int(first), int(count), int(key_bits), int(mul), int(add), int(mod), raw.ptr))
        k, _ = self.sort_count(raw, key_bits)
        del raw
        k = self.copy_of(k)
        if not counts:
            return k, None
        c = self.empty(k.n, np.uint64)
        self._check(self.lib.zk_synth_counts(self.h, int(seed), k.ptr, k.n, c.ptr))
        return k, c

    def copy_of(self, a):
        """A right-sized copy of a (possibly oversized or borrowed) device array."""
        out = self.empty(a.n, a.dtype)
        if a.n:
            self._check(self.lib.zk_copy(self.h, out.ptr, a.ptr, a.nbytes))
            self.sync()
        return out

    def hash_partition(self, kmers, counts, world, seed=0, out=None):
        """Stable split of a sorted table by the hash-range owner -> (kmers, counts | None, offsets[world + 1])."""
        if out is None:
            ok = self.empty(kmers.n, np.uint64)
            oc = self.empty(kmers.n, counts.dtype) if counts is not None else None
        else:
            ok, oc = out
        offs = (C.c_uint64 * (world + 1))()
        self._check(self.lib.zk_hash_partition(self.h, kmers.ptr, counts.ptr if counts is not None else None,
                                               counts.dtype.itemsize * 8 if counts is not None else 64, kmers.n, int(world), int(seed),
                                               ok.ptr, oc.ptr if oc is not None else None, offs))
        return ok, oc, [int(v) for v in offs]

    # ---- multi-GPU seam (zk_comm_*: RCCL bound by the library itself) -----------------------------------
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        rc = load().zk_comm_unique_id(buf)
        if rc != ZK_OK:
            raise ZotkError(rc, "zk_comm_unique_id failed (RCCL not loadable?)")
        return bytes(buf)

    def comm_init(self, world, rank, uid):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._check(self.lib.zk_comm_init(self.h, int(world), int(rank), buf))

    def comm_destroy(self):
        self._check(self.lib.zk_comm_destroy(self.h))

    def comm_info(self):
        w, r = C.c_int(1), C.c_int(0)
        self._check(self.lib.zk_comm_info(self.h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def all_to_all_v(self, send_ptr, send_off, send_cnt, recv_ptr, recv_off, recv_cnt, elem_bytes):
        W = len(send_cnt)
        arr = lambda v: (C.c_uint64 * W)(*[int(x) for x in v])
        self._check(self.lib.zk_all_to_all_v(self.h, send_ptr, arr(send_off), arr(send_cnt), recv_ptr, arr(recv_off), arr(recv_cnt),
                                             int(elem_bytes)))

    @staticmethod
    def comm_plan(world, rank, send_off, send_cnt, recv_off, recv_cnt, elem_bytes, chunk_bytes=0, self_loop=False):
        """The messages zk_all_to_all_v issues for this rank, as a list of dicts (host only: no GPU, no RCCL) -- zk_comm_plan."""
        class Op(C.Structure):
            _fields_ = [("recv", C.c_int32), ("peer", C.c_int32), ("round", C.c_uint64), ("offset", C.c_uint64), ("bytes", C.c_uint64)]
        arr = lambda v: (C.c_uint64 * world)(*[int(x) for x in v])
        a = (arr(send_off), arr(send_cnt), arr(recv_off), arr(recv_cnt))
        n = C.c_uint64(0)
        lib = load()
        rc = lib.zk_comm_plan(world, rank, a[0], a[1], a[2], a[3], int(elem_bytes), int(chunk_bytes), int(bool(self_loop)), None, 0, C.byref(n))
        if rc != ZK_OK:
            raise ZotkError(rc, "zk_comm_plan: bad argument")
        ops = (Op * max(n.value, 1))()
        lib.zk_comm_plan(world, rank, a[0], a[1], a[2], a[3], int(elem_bytes), int(chunk_bytes), int(bool(self_loop)), ops, n.value, C.byref(n))
        return [dict(recv=o.recv, peer=o.peer, round=o.round, offset=o.offset, bytes=o.bytes) for o in ops[:n.value]]

    def allreduce_u64(self, vals, op=0):
        a = np.ascontiguousarray(vals, dtype=np.uint64).copy()
        if a.size:
            self._check(self.lib.zk_allreduce_u64(self.h, a.ctypes.data_as(_pu64), a.size, int(op)))
        return a

    def stream_checksum(self, stream, K):
        s = (C.c_uint64 * 7)()
        self._check(self.lib.zk_stream_checksum(self.h, stream.ptr, stream.n, K, s))
        return tuple(int(v) for v in s[:3])

    def stream_acgt(self, stream, K):
        """acgt[x & 3] over every k-mer instance of the stream (commands/kmerize.py:492-493)"""
        s = (C.c_uint64 * 7)()
        self._check(self.lib.zk_stream_checksum(self.h, stream.ptr, stream.n, K, s))
        return [int(v) for v in s[3:7]]

    def capture_filter(self, stream, K, baits):
        out = self.empty(stream.n, np.uint8)
        nr, nk = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.zk_capture_filter(self.h, stream.ptr, stream.n, K, baits.ptr, baits.n, out.ptr, C.byref(nr), C.byref(nk)))
        return out, nr.value, nk.value
