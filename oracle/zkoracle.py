"""
ctypes/numpy face of the CPU oracle (oracle/zk_oracle.c) plus the few pieces of the
reference's hot path that are plain Python arithmetic on the host (the closed-form distance
measures and the container layout).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg -- never by anything under zotmer_amd/.  Parity status: pinned by
tests/golden/ (vectors captured from the reference; see tests/golden/README.md).

Reference file:line citations are relative to /root/reference.
"""
import ctypes as C
import json
import math
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


def build():
    """Compile oracle/zk_oracle.c with gcc (a no-op when the .so is newer than the source)."""
    so = os.path.join(_HERE, "libzkoracle.so")
    src = os.path.join(_HERE, "zk_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-shared", "-o", so, src])
    return so


class _Result(C.Structure):
    _fields_ = [("kmers", u64p), ("counts", u32p), ("n_unique", C.c_uint64),
                ("acgt", C.c_uint64 * 4), ("n_reads", C.c_uint64), ("n_kept", C.c_uint64),
                ("overflow", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.zo_rev.restype = C.c_uint64; L.zo_rev.argtypes = [C.c_uint64]
        L.zo_popcnt.restype = C.c_int; L.zo_popcnt.argtypes = [C.c_uint64]
        L.zo_ffs.restype = C.c_int; L.zo_ffs.argtypes = [C.c_uint64]
        L.zo_kmer.restype = C.c_uint64; L.zo_kmer.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        L.zo_render.restype = None; L.zo_render.argtypes = [C.c_int, C.c_uint64, C.c_char_p]
        L.zo_rc.restype = C.c_uint64; L.zo_rc.argtypes = [C.c_int, C.c_uint64]
        L.zo_ham.restype = C.c_int; L.zo_ham.argtypes = [C.c_uint64, C.c_uint64]
        L.zo_lcp.restype = C.c_int; L.zo_lcp.argtypes = [C.c_int, C.c_uint64, C.c_uint64]
        L.zo_fnv.restype = C.c_uint64; L.zo_fnv.argtypes = [C.c_uint64, C.c_uint64]
        L.zo_murmer.restype = C.c_uint64; L.zo_murmer.argtypes = [C.c_uint64, C.c_uint64]
        L.zo_can.restype = C.c_uint64; L.zo_can.argtypes = [C.c_int, C.c_uint64]
        L.zo_sub.restype = C.c_int; L.zo_sub.argtypes = [C.c_uint64, C.c_double, C.c_uint64]
        L.zo_kmers_list.restype = C.c_uint64
        L.zo_kmers_list.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.c_int, u64p, C.c_uint64]
        L.zo_radix_sort.restype = C.c_int; L.zo_radix_sort.argtypes = [C.c_int, u64p, C.c_uint64]
        L.zo_rle_merge.restype = C.c_uint64
        L.zo_rle_merge.argtypes = [u64p, u32p, C.c_uint64, u64p, C.c_uint64, u64p, u32p, C.POINTER(C.c_int)]
        L.zo_kmerize.restype = C.c_int
        L.zo_kmerize.argtypes = [C.c_int, C.c_char_p, u64p, C.c_uint64, C.c_int, C.c_double, C.c_uint64,
                                 u64p, C.c_uint64, C.c_uint64, C.POINTER(_Result)]
        L.zo_kmerize_free.restype = None; L.zo_kmerize_free.argtypes = [C.POINTER(_Result)]
        L.zo_hist.restype = C.c_uint64; L.zo_hist.argtypes = [u64p, C.c_uint64, u64p, u64p]
        L.zo_union_sum.restype = C.c_uint64
        L.zo_union_sum.argtypes = [u64p, u64p, C.c_uint64, u64p, u64p, C.c_uint64, u64p, u64p]
        L.zo_merge_n.restype = C.c_uint64
        L.zo_merge_n.argtypes = [C.c_int, C.c_int, C.POINTER(u64p), C.POINTER(u64p), u64p, u64p, u64p, u64p]
        L.zo_project_dedupe.restype = C.c_uint64
        L.zo_project_dedupe.argtypes = [u64p, C.c_uint64, C.c_int, u64p]
        L.zo_split.restype = None; L.zo_split.argtypes = [u64p, C.c_uint64, u64p, C.c_uint64, u64p]
        L.zo_trim.restype = C.c_uint64
        L.zo_trim.argtypes = [u64p, u64p, C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p]
        L.zo_project.restype = C.c_uint64
        L.zo_project.argtypes = [u64p, C.c_uint64, u64p, u64p, C.c_uint64, u64p, u64p]
        L.zo_sample_d.restype = C.c_uint64
        L.zo_sample_d.argtypes = [C.c_double, C.c_uint64, u64p, u64p, C.c_uint64, u64p, u64p]
        L.zo_codec64_encode.restype = C.c_int64; L.zo_codec64_encode.argtypes = [u64p, C.c_uint64, u64p]
        L.zo_codec64_decode.restype = C.c_int64; L.zo_codec64_decode.argtypes = [u64p, C.c_uint64, u64p]
        L.zo_delta.restype = None; L.zo_delta.argtypes = [u64p, C.c_uint64, u64p]
        L.zo_undelta.restype = None; L.zo_undelta.argtypes = [u64p, C.c_uint64, u64p]
        _LIB = L
    return _LIB


def _a64(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def _p64(a):
    return a.ctypes.data_as(u64p)


def _p32(a):
    return a.ctypes.data_as(u32p)


# ---- primitives ---------------------------------------------------------------------------

def rev(x): return lib().zo_rev(x)
def popcnt(x): return lib().zo_popcnt(x)
def ffs(x): return lib().zo_ffs(x)
def rc(k, x): return lib().zo_rc(k, x)
def ham(x, y): return lib().zo_ham(x, y)
def lcp(k, x, y): return lib().zo_lcp(k, x, y)
def fnv(x, s): return lib().zo_fnv(x, s)
def murmer(x, s): return lib().zo_murmer(x, s)
def can(k, x): return lib().zo_can(k, x)
def sub(s, p, x): return bool(lib().zo_sub(s, p, x))


def kmer(seq):
    ok = C.c_int(0)
    b = seq.encode() if isinstance(seq, str) else seq
    r = lib().zo_kmer(b, len(b), C.byref(ok))
    return r if ok.value else None


def render(k, x):
    buf = C.create_string_buffer(k + 1)
    lib().zo_render(k, x, buf)
    return buf.value.decode()


def kmers_list(k, seq, both=False):
    b = seq.encode() if isinstance(seq, str) else bytes(seq)
    cap = max(0, len(b) - k + 1) * (2 if both else 1)
    out = np.empty(max(cap, 1), dtype=np.uint64)
    n = lib().zo_kmers_list(k, b, len(b), int(both), _p64(out), cap)
    return out[:n].copy()


def radix_sort(bits, xs):
    a = _a64(xs).copy()
    if lib().zo_radix_sort(bits, _p64(a), len(a)):
        raise MemoryError
    return a


def rle_merge(xs, cs, ys):
    xs = _a64(xs); cs = np.ascontiguousarray(cs, dtype=np.uint32); ys = _a64(ys)
    zs = np.empty(len(xs) + len(ys), dtype=np.uint64)
    ss = np.empty(len(xs) + len(ys), dtype=np.uint32)
    ov = C.c_int(0)
    n = lib().zo_rle_merge(_p64(xs), _p32(cs), len(xs), _p64(ys), len(ys), _p64(zs), _p32(ss), C.byref(ov))
    if ov.value:
        raise OverflowError("count does not fit 32 bits")
    return zs[:n].copy(), ss[:n].copy()


def kmerize(K, reads, mode=0, p=0.0, seed=0, baits=None, flush_at=0):
    """In-memory `zot kmerize` over a list of read sequences (str/bytes).
    Returns dict(kmers u64[], counts u32[], acgt [4 ints], reads, kept)."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return kmerize_packed(K, b"".join(bs), offs, mode, p, seed, baits, flush_at)


def kmerize_packed(K, bases, offs, mode=0, p=0.0, seed=0, baits=None, flush_at=0):
    offs = _a64(offs)
    bt = _a64(baits if baits is not None else [])
    res = _Result()
    buf = bases if isinstance(bases, (bytes, bytearray)) else bytes(bases)
    rc_ = lib().zo_kmerize(K, buf, _p64(offs), len(offs) - 1, mode, float(p), int(seed),
                           _p64(bt), len(bt), int(flush_at), C.byref(res))
    if rc_:
        raise MemoryError
    n = res.n_unique
    out = dict(
        kmers=np.ctypeslib.as_array(res.kmers, shape=(n,)).copy() if n else np.empty(0, np.uint64),
        counts=np.ctypeslib.as_array(res.counts, shape=(n,)).copy() if n else np.empty(0, np.uint32),
        acgt=[int(v) for v in res.acgt], reads=int(res.n_reads), kept=int(res.n_kept),
        overflow=bool(res.overflow))
    lib().zo_kmerize_free(C.byref(res))
    return out


def hist(counts):
    c = _a64(counts)
    v = np.empty(max(len(c), 1), dtype=np.uint64)
    f = np.empty(max(len(c), 1), dtype=np.uint64)
    m = lib().zo_hist(_p64(c), len(c), _p64(v), _p64(f))
    return v[:m].copy(), f[:m].copy()


def union_sum(xs, xc, ys, yc):
    xs, xc, ys, yc = _a64(xs), _a64(xc), _a64(ys), _a64(yc)
    zs = np.empty(len(xs) + len(ys), dtype=np.uint64)
    zc = np.empty(len(xs) + len(ys), dtype=np.uint64)
    n = lib().zo_union_sum(_p64(xs), _p64(xc), len(xs), _p64(ys), _p64(yc), len(ys), _p64(zs), _p64(zc))
    return zs[:n].copy(), zc[:n].copy()


def merge_n(K, sets):
    """sets = [(kmers, counts), ...] -> (kmers, counts, acgt_weighted[4])."""
    k = len(sets)
    xs = [_a64(s[0]) for s in sets]
    xc = [_a64(s[1]) for s in sets]
    ns = np.array([len(x) for x in xs], dtype=np.uint64)
    px = (u64p * k)(*[_p64(x) for x in xs])
    pc = (u64p * k)(*[_p64(c) for c in xc])
    tot = int(ns.sum())
    zs = np.empty(max(tot, 1), dtype=np.uint64)
    zc = np.empty(max(tot, 1), dtype=np.uint64)
    ac = np.zeros(4, dtype=np.uint64)
    n = lib().zo_merge_n(K, k, px, pc, _p64(ns), _p64(zs), _p64(zc), _p64(ac))
    return zs[:n].copy(), zc[:n].copy(), [int(v) for v in ac]


def project_dedupe(xs, shift):
    xs = _a64(xs)
    out = np.empty(max(len(xs), 1), dtype=np.uint64)
    m = lib().zo_project_dedupe(_p64(xs), len(xs), shift, _p64(out))
    return out[:m].copy()


def split(xs, ys):
    xs, ys = _a64(xs), _a64(ys)
    abc = np.zeros(3, dtype=np.uint64)
    lib().zo_split(_p64(xs), len(xs), _p64(ys), len(ys), _p64(abc))
    return tuple(int(v) for v in abc)


def trim(xs, cs, lo, hi=0):
    xs, cs = _a64(xs), _a64(cs)
    ox = np.empty(max(len(xs), 1), dtype=np.uint64)
    oc = np.empty(max(len(xs), 1), dtype=np.uint64)
    m = lib().zo_trim(_p64(xs), _p64(cs), len(xs), lo, hi, _p64(ox), _p64(oc))
    return ox[:m].copy(), oc[:m].copy()


def project(xs, ys, yc):
    xs, ys, yc = _a64(xs), _a64(ys), _a64(yc)
    ok = np.empty(max(len(ys), 1), dtype=np.uint64); oc = np.empty(max(len(ys), 1), dtype=np.uint64)
    m = lib().zo_project(_p64(xs), len(xs), _p64(ys), _p64(yc), len(ys), _p64(ok), _p64(oc))
    return ok[:m].copy(), oc[:m].copy()


def sample_d(p, seed, ys, yc):
    ys, yc = _a64(ys), _a64(yc)
    ok = np.empty(max(len(ys), 1), dtype=np.uint64); oc = np.empty(max(len(ys), 1), dtype=np.uint64)
    m = lib().zo_sample_d(float(p), int(seed), _p64(ys), _p64(yc), len(ys), _p64(ok), _p64(oc))
    return ok[:m].copy(), oc[:m].copy()


# ---- zot jaccard statistics: zotmer/commands/jaccard.py:56-88, zotmer/library/stats.py:36-129 ----------

_LOG_SMALL_FAC = [math.log(math.factorial(n)) for n in range(25)]            # stats.py:45


def log_fac(n):                                                               # stats.py:77-83
    if n < len(_LOG_SMALL_FAC):
        return _LOG_SMALL_FAC[n]
    return n * math.log(n) - n + math.log(n * (1 + 4 * n * (1 + 2 * n))) / 6.0 + math.log(math.pi) / 2.0


def log_choose(n, k):                                                         # stats.py:121-128
    if k == 0 or k == n:
        return 0
    return log_fac(n) - (log_fac(n - k) + log_fac(k))


def log_add(a, b):                                                            # stats.py:85-92
    x, y = max(a, b), min(a, b)
    return x + math.log1p(math.exp(y - x))


def log_ix(x, m, n):                                                          # jaccard.py:56-71
    lx = math.log(x)
    j = m
    v = log_choose(n + j - 1, j)
    s = v + j * lx
    while True:
        j += 1
        v += math.log((n + j - 1.0) / j)
        t = v + j * lx
        u = log_add(s, t)
        if u == s:
            break
        s = u
    return n * math.log1p(-x) + s


def quant_beta(q, m, n):                                                      # jaccard.py:73-84
    lq = math.log(q)
    lo, hi = 1e-10, 1 - 1e-10
    while (hi - lo) > 1e-7:
        x = (hi + lo) / 2.0
        if log_ix(x, m, n) < lq:
            lo = x
        else:
            hi = x
    return lo


def jaccard_line(xn, yn, nx, ny, isec, p=None):
    """One output line of zot jaccard (jaccard.py:117-124,142-149)."""
    union = nx + ny - isec
    d = float(isec) / float(union)
    if p is None:
        return "%s\t%s\t%d\t%d\t%d\t%d\t%f" % (xn, yn, nx, ny, isec, union, d)
    pv = log_ix(p, isec + 1, (union - isec) + 1) / math.log(10)
    q05 = quant_beta(0.05, isec + 1, (union - isec) + 1)
    q95 = quant_beta(0.95, isec + 1, (union - isec) + 1)
    return "%s\t%s\t%d\t%d\t%d\t%d\t%f\t-%f\t+%f\t%f" % (xn, yn, nx, ny, isec, union, d, d - q05, q95 - d, pv)


def codec64_encode(xs):
    xs = _a64(xs)
    w = np.empty(max(len(xs), 1), dtype=np.uint64)
    n = lib().zo_codec64_encode(_p64(xs), len(xs), _p64(w))
    if n < 0:
        raise IndexError("value >= 2**60 has no codec64 code")
    return w[:n].copy()


def codec64_decode(words):
    w = _a64(words)
    n = lib().zo_codec64_decode(_p64(w), len(w), None)
    if n < 0:
        raise KeyError("codec64 tag with no width")
    out = np.empty(max(n, 1), dtype=np.uint64)
    lib().zo_codec64_decode(_p64(w), len(w), _p64(out))
    return out[:n].copy()


def delta(xs):
    xs = _a64(xs); d = np.empty_like(xs)
    lib().zo_delta(_p64(xs), len(xs), _p64(d))
    return d


def undelta(ds):
    ds = _a64(ds); x = np.empty_like(ds)
    lib().zo_undelta(_p64(ds), len(ds), _p64(x))
    return x


# ---- qualitative distance measures: zotmer/library/dist.py (set form) ----------------------
# Each keeps the reference's operation order so the doubles come out bit-identical.

def _f(x): return float(x)

QUAL_MEASURES = {
    # brayCurtis :40-41 and sorensen :209-210 share one formula
    "bray.curtis.qual": lambda a, b, c: _f(b + c) / _f(2 * a + b + c),
    # chord :67-68 and hellinger :93-94 share one formula
    "chord.qual": lambda a, b, c: math.sqrt(2 * (1 - a / math.sqrt((a + b) * (a + c)))),
    "hellinger.qual": lambda a, b, c: math.sqrt(2 * (1 - a / math.sqrt((a + b) * (a + c)))),
    # jaccard :112-113
    "jaccard.qual": lambda a, b, c: _f(b + c) / _f(a + b + c),
    # kulczynski :168-172
    "kulczynski.qual": lambda a, b, c: 1 - 0.5 * (_f(a) / (_f(a) + _f(b)) + _f(a) / (_f(a) + _f(c))),
    # ochiai :190-191
    "ochiai.qual": lambda a, b, c: 1 - a / math.sqrt((a + b) * (a + c)),
    # sorensen :209-210
    "sorensen.qual": lambda a, b, c: _f(b + c) / _f(2 * a + b + c),
    # whittaker :235-239
    "whittaker.qual": lambda a, b, c: 0.5 * (_f(b) / (_f(a) + _f(b)) + _f(c) / (_f(a) + _f(c))
                                              + abs(_f(a) / (_f(a) + _f(b)) - _f(a) / (_f(a) + _f(c)))),
}


# ---- container layout: zotmer/library/container/casket.py:219-234, zotmer/library/kmers.py --

def read_casket(path):
    """Return {member: bytes} (last version of each, casket.py:185) of a casket file."""
    with open(path, "rb") as f:
        blob = f.read()
    (z,) = struct.unpack("<Q", blob[-8:])
    toc = json.loads(blob[-8 - z:-8].decode())
    return {nm: blob[v[-1][0]:v[-1][0] + v[-1][1]] for nm, v in toc.items()}


def read_kmer_set(path):
    """Decode a `kmers` container: (meta dict, kmers u64[], counts u64[])."""
    m = read_casket(path)
    meta = json.loads(m["__meta__"].decode())
    km = undelta(codec64_decode(np.frombuffer(m[meta.get("kmers", "kmers")], dtype="<u8")))
    ct = codec64_decode(np.frombuffer(m[meta.get("counts", "counts")], dtype="<u8"))
    return meta, km, ct
