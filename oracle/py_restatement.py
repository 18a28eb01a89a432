"""
TEST / BASELINE INFRASTRUCTURE -- never imported by the product (zotmer_amd/).

A pure-Python restatement of the reference's `zot kmerize` CPU path, with the reference's algorithmic structure, used
as the "reference CPU path" figure that BASELINE.md section 4 / SURVEY.md section 8(d) prescribe next to the GPU numbers
(the reference's own source cannot travel to the GPU box, and there is no pypy / python2 anywhere):

  per-read sliding window, both strands     basics.kmersList            zotmer/library/basics.py:303-347
  acgt[x & 3] per emitted k-mer              kmerize.main                zotmer/commands/kmerize.py:492-493
  buffer -> bucket by top byte(s) -> sort    misc.radix_sort             zotmer/library/misc.py:400-424
  run-length count fused with 2-way merge    kmerize.merge               zotmer/commands/kmerize.py:41-132
  accumulate / flush policy                  KmerAccumulator2            zotmer/commands/kmerize.py:370-437
  delta + codec64 greedy word packing        files.delta, codec64.encode zotmer/library/files.py:85-98, codec64.py:42-120

Pinned: tests/test_oracle_golden.py runs it on BASELINE config 1 (10 000 x 150 bp, K = 25, seed 20261004) and compares
the sha256 of both encoded streams, hist, acgt and reads with tests/golden/config1_digest.json, which the reference itself
produced (tests/golden/make_golden.py).  Single-threaded like the reference; everything is Python ints and lists.
"""
import hashlib
import struct

_CODE = [None] * 256
for _i, _ch in enumerate("ACGT"):
    _CODE[ord(_ch)] = _i
    _CODE[ord(_ch.lower())] = _i
_CODE[ord("U")] = _CODE[ord("u")] = 3


def windows_both_strands(k, seq):
    """basics.kmersList(k, seq, True): x then rc(x) for every window of k valid bases; any other byte restarts."""
    out = []
    top = 2 * (k - 1)
    mask = (1 << (2 * k)) - 1
    fwd = rev = run = 0
    emit = out.append
    for ch in seq:
        b = _CODE[ord(ch)] if ord(ch) < 256 else None
        if b is None:
            fwd = rev = run = 0
            continue
        fwd = ((fwd << 2) | b) & mask
        rev = (rev >> 2) | ((3 - b) << top)
        run += 1
        if run >= k:
            emit(fwd)
            emit(rev)
    return out


def bucket_sort(bits, xs, small=16384):
    """misc.radix_sort: up to two levels of 256-way bucketing on the top bytes, then list.sort per bucket."""
    def rec(depth, part, into):
        if depth >= 2 or (depth + 1) * 8 >= bits or len(part) <= small:
            part.sort()
            into.extend(part)
            return
        shift = bits - (depth + 1) * 8
        buckets = [[] for _ in range(256)]
        for x in part:
            buckets[(x >> shift) & 255].append(x)
        for b in buckets:
            rec(depth + 1, b, into)
    if len(xs) <= small:
        xs.sort()
        return xs
    out = []
    rec(0, xs, out)
    return out


def rle_merge(tk, tc, ys):
    """kmerize.merge: the sorted raw list ys is run-length counted on the fly and merged into the sorted distinct table
    (tk, tc); equal k-mers add."""
    ok, oc = [], []
    i, n = 0, len(tk)
    j, m = 0, len(ys)
    while j < m:
        y = ys[j]
        e = j + 1
        while e < m and ys[e] == y:
            e += 1
        while i < n and tk[i] < y:
            ok.append(tk[i])
            oc.append(tc[i])
            i += 1
        if i < n and tk[i] == y:
            ok.append(y)
            oc.append(tc[i] + (e - j))
            i += 1
        else:
            ok.append(y)
            oc.append(e - j)
        j = e
    while i < n:
        ok.append(tk[i])
        oc.append(tc[i])
        i += 1
    return ok, oc


# widths of the codec64 words: n values of 60 // n bits, only where that divides evenly (codec64.py:26-40)
_FIT = [(0, 64)] * 61
for _n in range(1, 61):
    _FIT[_n] = (60 // _n, _n) if 60 % _n == 0 else _FIT[_n - 1]


def codec64_words(values):
    """codec64.encode: greedy -- keep appending while the widest pending value still fits the next word shape."""
    words = []
    pend, widest = [], 0
    for x in values:
        w = max(x.bit_length(), widest)
        n = len(pend)
        if n == 60 or w > _FIT[n + 1][0] or n >= _FIT[n + 1][1]:
            bits, take = _FIT[n]
            v = 0
            for t in range(take - 1, -1, -1):
                v = (v << bits) | pend[t]
            words.append((v << 4) | take)
            del pend[:take]
            widest = max([p.bit_length() for p in pend]) if pend else 0
            w = max(x.bit_length(), widest)
        pend.append(x)
        widest = w
    if pend:
        bits, take = _FIT[len(pend)]
        v = 0
        for t in range(take - 1, -1, -1):
            v = (v << bits) | pend[t]
        words.append((v << 4) | take)
    return words


def kmerize(k, reads, flush_at=128 * 1024 * 1024):
    """`zot kmerize k out reads` in memory: dict(kmers, counts, hist, acgt (normalised), reads, instances, kmers_bytes,
    counts_bytes) where the two byte strings are the codec64 members the reference writes."""
    acgt = [0, 0, 0, 0]
    tk, tc, buf = [], [], []
    n_reads = 0
    for r in reads:
        xs = windows_both_strands(k, r)
        for x in xs:
            acgt[x & 3] += 1
        buf.extend(xs)
        n_reads += 1
        if len(buf) > len(tk) and len(buf) > flush_at:
            tk, tc = rle_merge(tk, tc, bucket_sort(2 * k, buf))
            buf = []
    if buf:
        tk, tc = rle_merge(tk, tc, bucket_sort(2 * k, buf))
    hist = {}
    for c in tc:
        hist[c] = hist.get(c, 0) + 1
    deltas, prev = [], 0
    for x in tk:
        deltas.append(x - prev)
        prev = x
    kw, cw = codec64_words(deltas), codec64_words(tc)
    total = float(sum(acgt))
    return dict(kmers=tk, counts=tc, hist=hist, acgt=[a / total for a in acgt] if total else acgt, reads=n_reads,
                instances=sum(acgt), kmers_bytes=struct.pack("<%dQ" % len(kw), *kw), counts_bytes=struct.pack("<%dQ" % len(cw), *cw))


def digest(res):
    return dict(sha256_kmers=hashlib.sha256(res["kmers_bytes"]).hexdigest(), sha256_counts=hashlib.sha256(res["counts_bytes"]).hexdigest(),
                len_raw_kmers=len(res["kmers_bytes"]), len_raw_counts=len(res["counts_bytes"]))
