"""
GPU parity: the HIP path, called through the C-ABI (zotmer_amd/native.py -> libzotk.so), against
  (a) the golden vectors captured from the reference (tests/golden), and
  (b) the CPU oracle (oracle/zk_oracle.c) on the same seeded inputs,
bit-exact throughout (integer / byte work: no tolerance anywhere).
"""
import hashlib

import numpy as np
import pytest

from oracle import zkoracle as zo
from tests import _golden as G
from zotmer_amd import native, synth

pytestmark = pytest.mark.gpu
DEFAULT_SORT_VARIANT = 3       # zk_ctx defaults (internal.hpp)
DEFAULT_PAIRS_VARIANT = 7


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


def stream_of(reads):
    return ("".join(r + "\n" for r in reads)).encode()


# ---- synthetic input ------------------------------------------------------------------------------

@pytest.mark.parametrize("kw", [dict(genome=0, n_thr=synth.frac32(0.01)),
                                dict(genome=5000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.0005)),
                                dict(genome=150)])
def test_synth_device_matches_numpy(ctx, kw):
    for first, count, L in ((0, 777, 150), (12345, 300, 37)):
        if kw.get("genome") and kw["genome"] < L:
            continue
        want = synth.base_stream(synth.DEFAULT_SEED, first, count, L, **kw)
        got = ctx.synth_reads(synth.DEFAULT_SEED, first, count, L, **kw).to_host()
        assert np.array_equal(got, want)


def test_can_golden_and_oracle(ctx):
    """basics.can on the device against the reference's own values (tests/golden/primitives.json) and the oracle"""
    P = G.load_json("primitives")
    for K in sorted({c[0] for c in P["can"]}):
        rows = [c for c in P["can"] if c[0] == K]
        got = ctx.can(K, ctx.upload(np.array([c[1] for c in rows], dtype=np.uint64))).to_host()
        assert [int(v) for v in got] == [c[2] for c in rows]
    rng = np.random.default_rng(9)
    for K in (1, 16, 25, 31, 32):
        x = rng.integers(0, 1 << 63, size=5000, dtype=np.uint64) & np.uint64((1 << (2 * K)) - 1 if K < 32 else (1 << 64) - 1)
        assert [int(v) for v in ctx.can(K, ctx.upload(x)).to_host()] == [zo.can(K, int(v)) for v in x]


# ---- K1 encode --------------------------------------------------------------------------------------

def test_encode_golden_kmersList(ctx):
    P = G.load_json("primitives")
    for c in P["kmersList"]:
        if c["k"] > 32:
            continue
        d = ctx.upload_stream(c["seq"].encode() + b"\n")
        out, _ = ctx.encode(d, c["k"], c["both"])
        assert [int(x) for x in out.to_host()] == c["out"], (c["k"], c["seq"], c["both"])


@pytest.mark.parametrize("K", [1, 5, 16, 25, 31, 32])
def test_encode_many_reads_vs_oracle(ctx, K):
    reads = synth.read_strings(7, 0, 3000, 97, genome=0, n_thr=synth.frac32(0.02))
    reads += ["", "ACGT", "acgtnACGTUuuuacgtacgtagcatgcatgcatcgatcgatgcatgcatgcatgcatgcatgca", "N" * 40, "A" * 100]
    want = np.concatenate([zo.kmers_list(K, r, True) for r in reads])
    d = ctx.upload_stream(stream_of(reads))
    out, acgt = ctx.encode(d, K, True)
    got = out.to_host()
    assert np.array_equal(got, want)
    assert acgt == [int(np.sum((want & np.uint64(3)) == np.uint64(b))) for b in range(4)]
    out1, _ = ctx.encode(d, K, False)
    assert np.array_equal(out1.to_host(), want[0::2])


def test_pack_reads(ctx):
    reads = synth.read_strings(9, 0, 500, 61, genome=0) + ["", "ACGT", "T" * 200]
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    s = ctx.pack_reads(ctx.upload(bases), ctx.upload(offs))
    assert s.to_host().tobytes() == stream_of(reads)


# ---- K3 sort ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n", [0, 1, 63, 64, 100, 8191, 8192, 8193, 100003, 1 << 20])
@pytest.mark.parametrize("bits", [2, 9, 50, 64])
def test_sort_keys(ctx, n, bits):
    rng = np.random.default_rng(n * 131 + bits)
    x = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    if bits < 64:
        x &= np.uint64((1 << bits) - 1)
    got = ctx.sort_keys(ctx.upload(x), bits).to_host()
    assert np.array_equal(got, np.sort(x))


@pytest.mark.parametrize("variant", range(8))
def test_sort_keys_every_geometry(ctx, variant):
    """every instantiated tile geometry / look-back scheme (zk_tune): same result, stable for pairs (7: the pipeline with a payload,
    pairs only -- the keys then take the default geometry)"""
    try:
        ctx.tune(sort_variant=variant if variant < 7 else DEFAULT_SORT_VARIANT, pairs_variant=variant)
        for n in (1, 8192, 8192 * 33 + 7, 8192 * 200 + 4097):          # 1 tile; > one segment; several segments
            rng = np.random.default_rng(n + variant)
            x = rng.integers(0, 1 << 50, size=n, dtype=np.uint64)
            x[n // 3: n // 3 + min(n // 4, 40000)] = x[0]                # a skewed digit on top
            assert np.array_equal(ctx.sort_keys(ctx.upload(x), 50).to_host(), np.sort(x))
            k = (x >> np.uint64(38)) << np.uint64(20)
            v = np.arange(n, dtype=np.uint32)
            dk, dv = ctx.sort_pairs(ctx.upload(k), ctx.upload(v), 50)
            order = np.argsort(k, kind="stable")
            assert np.array_equal(dk.to_host(), k[order]) and np.array_equal(dv.to_host(), v[order])
            if variant == 3:          # the default geometry hands its array passes to the 16 K-key tiles (6): its own 8 K-key tiles as well
                try:
                    ctx.tune(wide_tiles=0)
                    assert np.array_equal(ctx.sort_keys(ctx.upload(x), 50).to_host(), np.sort(x))
                finally:
                    ctx.tune(wide_tiles=1)
    finally:
        ctx.tune(sort_variant=DEFAULT_SORT_VARIANT, pairs_variant=DEFAULT_PAIRS_VARIANT)


@pytest.mark.parametrize("group", [1, 8, 32])
def test_sort_keys_xcd_grouped_tile_order(ctx, group):
    """zk_tune(ZK_TUNE_XCD_GROUP): runs of tiles handed to one XCD, work stealing at the tail -- off by default,
    the result must not depend on it (including inputs smaller than one run and one-tile inputs)"""
    try:
        ctx.tune(xcd_group=group)
        for n in (1, 5000, 8192 * 9 + 1, 8192 * 300 + 77):
            rng = np.random.default_rng(n + group)
            x = rng.integers(0, 1 << 50, size=n, dtype=np.uint64)
            assert np.array_equal(ctx.sort_keys(ctx.upload(x), 50).to_host(), np.sort(x))
        reads = synth.read_strings(9, 0, 3000, 150, genome=20000, sub_thr=synth.frac32(0.01), n_thr=synth.frac32(0.001))
        want = zo.kmerize(25, reads)
        k, c, _ = ctx.kmerize(ctx.upload_stream(stream_of(reads)), 25)
        assert np.array_equal(k.to_host(), want["kmers"]) and np.array_equal(c.to_host(), want["counts"])
    finally:
        ctx.tune(xcd_group=0)


def test_sort_keys_adversarial(ctx):
    n = 300001
    for x in (np.full(n, 0x23c48f123c48f, dtype=np.uint64),
              np.arange(n, dtype=np.uint64), np.arange(n, dtype=np.uint64)[::-1].copy(),
              np.repeat(np.arange(17, dtype=np.uint64) << np.uint64(40), n // 17)):
        got = ctx.sort_keys(ctx.upload(x), 50).to_host()
        assert np.array_equal(got, np.sort(x))


def test_sort_keys_golden_radix_sort(ctx):
    g = G.load_json("primitives")["radix_sort"]
    xs = np.load(G.GOLD + "/radix_sort_in.npz")["xs"]
    got = ctx.sort_keys(ctx.upload(xs), g["bits"]).to_host()
    assert hashlib.sha256(got.astype("<u8").tobytes()).hexdigest() == g["sha256_out"]


@pytest.mark.parametrize("n", [1, 5000, 8192 * 3 + 5, 500000])
def test_sort_pairs_is_stable(ctx, n):
    rng = np.random.default_rng(n)
    k = rng.integers(0, 1 << 12, size=n, dtype=np.uint64) << np.uint64(20)      # many duplicates
    v = np.arange(n, dtype=np.uint32)
    dk, dv = ctx.sort_pairs(ctx.upload(k), ctx.upload(v), 50)
    order = np.argsort(k, kind="stable")
    assert np.array_equal(dk.to_host(), k[order])
    assert np.array_equal(dv.to_host(), v[order])


def _tile_sort_inputs(rng, n, bits):
    """inputs of the sort that is finished tile by tile in LDS (tilesort.hip): spread evenly, every key about three times, a few
    crowded stretches of the key space (blocks of equal top bits of up to ~1000 keys: the tile's groups are then uneven), one block
    too long for a tile (the sort must notice and take the long way)"""
    mask = np.uint64((1 << bits) - 1)
    top = np.uint64(bits - 18 if bits > 27 else 9)
    uni = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    yield "uniform", uni & mask
    yield "triples", (rng.integers(0, max(1, n // 3), size=n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) & mask
    x = uni & mask
    m = min(n // 2, 40 * 900)
    pre = rng.integers(0, 1 << 18, size=40, dtype=np.uint64) << top
    x[:m] = (np.repeat(pre, 900)[:m] | (x[:m] & ((np.uint64(1) << top) - np.uint64(1)))) & mask
    yield "crowded", x
    y = x.copy()
    y[n // 2: n // 2 + 1500] = (y[n // 2] >> top << top) | (y[n // 2: n // 2 + 1500] & np.uint64((1 << min(16, int(top))) - 1))
    yield "too_long", y


@pytest.mark.parametrize("bits", [30, 50, 62, 64])
@pytest.mark.parametrize("n", [65536, 65536 + 7168 + 3, 2_500_001])
def test_sort_finished_in_tiles(ctx, n, bits):
    """zk_sort_keys / zk_sort_pairs on arrays large enough for the tile sort (LSD passes over the top bits, the rest in LDS) against
    numpy's sort (library/misc.py:400-424); the pairs stay in their order of arrival when keys are equal; and the same with the
    tile sort switched off"""
    rng = np.random.default_rng(n * 7 + bits)
    for name, x in _tile_sort_inputs(rng, n, bits):
        want = np.sort(x)
        assert np.array_equal(ctx.sort_keys(ctx.upload(x), bits).to_host(), want), name
        v = np.arange(n, dtype=np.uint32)
        dk, dv = ctx.sort_pairs(ctx.upload(x), ctx.upload(v), bits)
        order = np.argsort(x, kind="stable")
        assert np.array_equal(dk.to_host(), want) and np.array_equal(dv.to_host(), v[order]), name
    try:
        ctx.tune(tile_sort=0)
        assert np.array_equal(ctx.sort_keys(ctx.upload(x), bits).to_host(), np.sort(x))
    finally:
        ctx.tune(tile_sort=1)


# ---- K4 run-length count ---------------------------------------------------------------------------------

@pytest.mark.parametrize("in_place", [False, True])
def test_rle(ctx, in_place):
    rng = np.random.default_rng(5)
    vals = np.sort(rng.choice(np.arange(1, 10 ** 7, dtype=np.uint64), size=30000, replace=False))
    reps = rng.integers(1, 6, size=len(vals))
    reps[100] = 30000           # a run across several 8192-element tiles
    reps[101] = 8192
    reps[5000] = 8193
    reps[-1] = 5000             # ... and one that ends the array
    x = np.repeat(vals, reps)
    u, c = ctx.rle(ctx.upload(x), in_place=in_place)
    assert np.array_equal(u.to_host(), vals)
    assert np.array_equal(c.to_host(), reps.astype(np.uint32))
    # one long run only, and the empty array
    u, c = ctx.rle(ctx.upload(np.full(100000, 42, dtype=np.uint64)))
    assert u.to_host().tolist() == [42] and c.to_host().tolist() == [100000]
    u, c = ctx.rle(ctx.upload(np.empty(0, dtype=np.uint64)))
    assert u.n == 0 and c.n == 0


def test_sort_count_vs_oracle(ctx):
    reads = synth.read_strings(11, 0, 4000, 150, genome=30000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.0005))
    inst = np.concatenate([zo.kmers_list(25, r, True) for r in reads])
    u, c = ctx.sort_count(ctx.upload(inst), 50)
    zs, ss = zo.rle_merge([], [], zo.radix_sort(50, inst))
    assert np.array_equal(u.to_host(), zs) and np.array_equal(c.to_host(), ss)


# ---- the kmerize batch -----------------------------------------------------------------------------------------

def _check_kmerize(ctx, K, reads, km, ct, meta, flags=native.KMERIZE_CANONICAL, **kw):
    d = ctx.upload_stream(stream_of(reads))
    k, c, st = ctx.kmerize(d, K, flags, **kw)
    assert np.array_equal(k.to_host(), km)
    assert np.array_equal(c.to_host().astype(np.uint64), ct)
    tot = float(sum(st.acgt))
    assert [v / tot for v in st.acgt] == meta["acgt"]                      # kmerize.py:554-555
    assert {str(a): b for a, b in ctx.hist(c).items()} == meta["hist"]     # kmerize.py:543-545
    return st


@pytest.mark.parametrize("name", G.KMERIZE_SYNTH_CASES)
@pytest.mark.parametrize("flags", [native.KMERIZE_CANONICAL, native.KMERIZE_BOTH])
def test_kmerize_golden(ctx, name, flags):
    info, km, ct, _, _ = G.load_case(name)
    st = _check_kmerize(ctx, info["K"], G.synth_reads(info), km, ct, info["meta"], flags)
    assert st.n_instances == info["sum_counts"] and st.n_unique == info["n"]


def test_kmerize_golden_edge_inputs(ctx):
    info, km, ct, _, _ = G.load_case("g9_edge_fastq")
    fq = G.fastq_seqs(info["fastq"])
    _check_kmerize(ctx, 25, fq, km, ct, info["meta"])
    info2, km2, ct2, _, _ = G.load_case("g9_edge_fasta")
    fa = G.fasta_seqs(info2["fasta"])
    _check_kmerize(ctx, 25, fa, km2, ct2, info2["meta"])
    info3, km3, ct3, _, _ = G.load_case("g9_two_files")
    _check_kmerize(ctx, 25, fq + fa, km3, ct3, info3["meta"])


def test_kmerize_golden_subsample(ctx):
    info, km, ct, _, _ = G.load_case("g10_kmerize_D0.8_S3")
    for flags in (native.KMERIZE_CANONICAL, native.KMERIZE_BOTH):
        _check_kmerize(ctx, 25, G.synth_reads(info), km, ct, info["meta"], flags | native.KMERIZE_SUBSAMPLE,
                       p=info["D"], seed=info["S"])


def test_subsample_golden_sub(ctx):
    P = G.load_json("primitives")
    for s in (0, 3):
        for p in (0.0, 0.05, 0.8, 1.0, 4.0, 7.99, 8.0, 9.0):
            rows = [r for r in P["sub"] if r[0] == s and r[1] == p]
            xs = np.array([r[2] for r in rows], dtype=np.uint64)
            got = ctx.subsample(ctx.upload(xs), s, p).to_host()
            assert got.tolist() == [r[2] for r in rows if r[3]]


def test_kmerize_config1_digest(ctx):
    g = G.load_json("config1_digest")
    d = ctx.upload_stream(synth.base_stream(**g["synth"]))
    # the device generator gives the same stream
    assert np.array_equal(ctx.synth_reads(**{k: v for k, v in g["synth"].items()}).to_host(), d.to_host())
    k, c, st = ctx.kmerize(d, g["K"])
    kh, ch = k.to_host(), c.to_host().astype(np.uint64)
    assert len(kh) == g["n"] and int(ch.sum()) == g["sum_counts"] == st.n_instances
    assert hashlib.sha256(kh.astype("<u8").tobytes()).hexdigest() == g["sha256_kmers"]
    assert hashlib.sha256(ch.astype("<u8").tobytes()).hexdigest() == g["sha256_counts"]
    assert {str(a): b for a, b in ctx.hist(c).items()} == g["meta"]["hist"]
    tot = float(sum(st.acgt))
    assert [v / tot for v in st.acgt] == g["meta"]["acgt"]


def test_kmerize_empty_and_tiny(ctx):
    for reads in ([], [""], ["ACGT"], ["N" * 100], ["A" * 25], ["ACGTACGTACGTACGTACGTACGTAC"]):
        want = zo.kmerize(25, reads)
        d = ctx.upload_stream(stream_of(reads)) if reads else ctx.empty(0, np.uint8)
        for flags in (native.KMERIZE_CANONICAL, native.KMERIZE_BOTH):
            k, c, st = ctx.kmerize(d, 25, flags)
            assert np.array_equal(k.to_host(), want["kmers"]) and np.array_equal(c.to_host(), want["counts"])
            assert list(st.acgt) == want["acgt"]


@pytest.mark.parametrize("K", [13, 25, 31, 32])
def test_kmerize_short_sort_paths_agree(ctx, K):
    """zk_kmerize can sort only the top ~log2(n)+3 bits and finish in the mirror stage (opt-in); the
    full sort, the short sort and the short sort with a side list too small to hold anything (forced
    fall-back) must give the same arrays, also on inputs built to crowd the prefix groups."""
    rng = np.random.default_rng(K)
    low = ["".join(rng.choice(list("AC"), size=120)) for _ in range(1500)]            # low complexity: shared prefixes
    poly = ["A" * 60 + "".join(rng.choice(list("ACGT"), size=40)) for _ in range(1500)]
    reads = synth.read_strings(5, 0, 2500, 150, genome=9000, sub_thr=synth.frac32(0.01), n_thr=synth.frac32(0.001)) + low + poly
    reads += ["ACGT" * 40, "T" * 150, "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTA" * 3] * 50
    want = zo.kmerize(K, reads)
    d = ctx.upload_stream(stream_of(reads))
    try:
        for kw in (dict(short_sort=0), dict(short_sort=1, side_div=8), dict(short_sort=1, side_div=1), dict(short_sort=1, side_div=10 ** 9)):
            ctx.tune(**kw)
            k, c, st = ctx.kmerize(d, K)
            assert np.array_equal(k.to_host(), want["kmers"]), kw
            assert np.array_equal(c.to_host(), want["counts"]), kw
            assert list(st.acgt) == want["acgt"]
    finally:
        ctx.tune(short_sort=0, side_div=8)


@pytest.mark.parametrize("K", [9, 14, 25, 26, 27, 31, 32])
def test_kmerize_early_collapse_paths_agree(ctx, K):
    """zk_kmerize (canonical) counts runs after the passes over the low bits and finishes the sort on (k-mer, count) pairs when a
    sample says the reads repeat their k-mers; otherwise the keys finish the sort.  Both ways, and with the collapse switched
    off, the arrays must be the oracle's -- also when distinct k-mers share all their low bits, so that their copies interleave
    and the final reduce has to add up split runs.  early_collapse = 1 counts in LDS hash tables over blocks of equal low bits
    (dedupe_kernel; counts beyond the packed field go out as several words), 3 inside the tile-local ranking of the last low
    digit (collapse_kernel: K <= 27, where a pair packs into one word; K = 26, 27 have fewer than 14 spare bits, so long runs
    are cut every 512 slots), 2 as a pass of its own."""
    rng = np.random.default_rng(100 + K)
    deep = synth.read_strings(11, 0, 6000, 150, genome=12000, sub_thr=synth.frac32(0.004), n_thr=synth.frac32(0.001))   # ~60x: collapses
    flat = synth.read_strings(12, 0, 3000, 150, genome=0)                                                              # no repeats: refused
    # k-mers that differ only in their FIRST bases (high bits) and share every low bit, many copies each, interleaved
    tail = "".join(rng.choice(list("ACGT"), size=40))
    shared = [h + tail for h in ("A", "C", "G", "T", "AC", "GT", "TTA", "CAG")] * 400
    rng.shuffle(shared)
    # counts beyond the spare bits of a packed (k-mer << s | count) word (K = 25: 14 bits): the packed forms must step aside
    heavy = deep[:1500] + ["A" * 150] * 400 + ["AC" * 75] * 300
    # one stretch of the key space with more distinct k-mers than an LDS table of the block dedupe holds (8000 one-window reads that
    # all start with AAAAA) in an input that otherwise repeats its k-mers: that block alone is counted by sorting
    dense = deep + ["AAAAA" + "".join(rng.choice(list("ACGT"), size=K - 5)) for _ in range(8000)] if K >= 12 else deep
    for name, reads in (("deep", deep), ("flat", flat), ("shared_low_bits", shared + deep[:500]), ("mixed", deep[:2000] + flat[:2000] + shared),
                        ("heavy_counts", heavy), ("dense_block", dense)):
        want = zo.kmerize(K, reads)
        d = ctx.upload_stream(stream_of(reads))
        try:
            for on, packed in ((1, 1), (3, 1), (2, 1), (1, 0), (0, 1), (0, 0)):
                ctx.tune(early_collapse=on, packed_pairs=packed)
                k, c, st = ctx.kmerize(d, K)
                assert np.array_equal(k.to_host(), want["kmers"]), (name, on, packed)
                assert np.array_equal(c.to_host(), want["counts"]), (name, on, packed)
                assert list(st.acgt) == want["acgt"] and st.n_unique == len(want["kmers"])
            # reads that do not repeat (and the mirrored pairs of K >= 28) finish their sort tile by tile in LDS (tilesort.hip): the
            # same arrays without it, and the counted canonical list both ways
            for ts in (0, 1):
                ctx.tune(early_collapse=1, packed_pairs=1, tile_sort=ts)
                k, c, st = ctx.kmerize(d, K)
                assert np.array_equal(k.to_host(), want["kmers"]) and np.array_equal(c.to_host(), want["counts"]), (name, "tile_sort", ts)
                ck, cc, _ = ctx.kmerize(d, K, native.KMERIZE_CANONICAL_ONLY)
                ek, ec = ctx.mirror_expand(ck, cc, K)
                assert np.array_equal(ek.to_host(), want["kmers"]) and np.array_equal(ec.to_host(), want["counts"]), (name, "canonical", ts)
        finally:
            ctx.tune(early_collapse=1, packed_pairs=1, tile_sort=1)


@pytest.mark.parametrize("K", [16, 20, 25, 27])
def test_kmerize_forced_block_dedupe(ctx, K):
    """The plan of a 50 M-read batch -- two passes over the top 18 bits, the second one writing 32-bit tags where the bits below fit
    (K <= 25), the blocks of equal top bits counted in LDS tables -- forced (ZK_TUNE_DEDUPE_BITS) on inputs small enough for the
    oracle, where nothing else reaches it: tags against whole keys (tag_words), dedupe2_kernel (two workgroups per CU; with and
    without the plain read before the swap) against dedupe_kernel alone, the second chance of the blocks dedupe2_kernel declines
    (dedupe_limit lowered so that ordinary blocks count as too large), a block with more distinct keys than any table holds (sorted
    by the host, from tags), counts beyond the packed field, and an input without repeats (declined after the passes: the keys are
    made again from their tags and sorted the long way).  18 block bits on ~1 M keys: blocks of a few keys, many of them empty."""
    rng = np.random.default_rng(300 + K)
    deep = synth.read_strings(31, 0, 6000, 150, genome=12000, sub_thr=synth.frac32(0.004), n_thr=synth.frac32(0.001))
    flat = synth.read_strings(32, 0, 3000, 150, genome=0)
    # 14 000 distinct k-mers that share their first nine bases (one block of the 18-bit plan), in an input that repeats its k-mers
    dense = deep + ["AAAAAAAAA" + "".join(rng.choice(list("ACGT"), size=K - 9)) for _ in range(14000)]
    heavy = deep[:1500] + ["A" * 150] * 400 + ["AC" * 75] * 300
    # one block with 70 000 keys (more than a 16-bit count may see) of a few hundred distinct k-mers
    big = deep[:1000] + ["AAAAAAAAA" + "".join(rng.choice(list("ACGT"), size=K - 9)) for _ in range(300)] * 240
    try:
        for name, reads in (("deep", deep), ("flat", flat), ("mixed", deep[:2000] + flat[:2000]), ("dense_block", dense), ("heavy_counts", heavy),
                            ("big_block", big)):
            want = zo.kmerize(K, reads)
            d = ctx.upload_stream(stream_of(reads))
            # tag_pass: the tags written by tag_pass.hip's pass over static segments (pass 0 leaves two arrays), or by the look-back pipeline
            # tag_words 2: ... which takes the places inside a digit's run from LDS adds wherever a tile holds keys of one bucket only
            # (big_block: nine such tiles)
            # (wide: the pass on 16 K-key tiles, the default, or on 8 K-key tiles)
            for tag_words, variant, limit, tag_pass, wide in ((1, 0, 65536, 1, 1), (1, 0, 65536, 0, 1), (0, 0, 65536, 1, 1), (1, 2, 65536, 1, 1), (1, -1, 65536, 0, 1),
                                                             (1, -1, 65536, 1, 1), (1, 0, 6, 1, 1), (0, 2, 6, 0, 1), (2, 0, 65536, 0, 1), (2, -1, 65536, 0, 1),
                                                             (2, 0, 65536, 0, 0), (1, 0, 65536, 0, 0), (0, 0, 65536, 0, 0)):
                ctx.tune(dedupe_bits=18, tag_words=tag_words, dedupe_variant=variant, dedupe_limit=limit, tag_pass=tag_pass, wide_tiles=wide)
                k, c, st = ctx.kmerize(d, K)
                assert np.array_equal(k.to_host(), want["kmers"]), (name, tag_words, variant, limit, tag_pass, wide)
                assert np.array_equal(c.to_host(), want["counts"]), (name, tag_words, variant, limit, tag_pass, wide)
                assert list(st.acgt) == want["acgt"] and st.n_unique == len(want["kmers"])
    finally:
        ctx.tune(dedupe_bits=0, tag_words=native.DEFAULT_TAG_WORDS, dedupe_variant=0, dedupe_limit=65536, tag_pass=0, wide_tiles=1)


@pytest.mark.parametrize("K", [4, 12, 24, 25, 31, 32])
def test_kmerize_canonical_only_and_mirror_expand(ctx, K):
    """zk_kmerize(ZK_KMERIZE_CANONICAL_ONLY) = the counted list of c = min(x, rc x); zk_mirror_expand of it = zk_kmerize.
    Even K has palindromes (counted twice per window in the both-strand table, once in the canonical list)."""
    reads = synth.read_strings(21, 0, 5000, 120, genome=15000, sub_thr=synth.frac32(0.004), n_thr=synth.frac32(0.002)) + ["ACGT" * 25, "AATT" * 20, "GC" * 50]
    want = zo.kmerize(K, reads)
    wk, wc = want["kmers"], want["counts"]
    rc = np.array([zo.rc(K, int(x)) for x in wk], dtype=np.uint64)
    keep = wk <= rc
    ck_want, cc_want = wk[keep], wc[keep].astype(np.uint64)
    cc_want[wk[keep] == rc[keep]] //= 2
    d = ctx.upload_stream(stream_of(reads))
    for collapse in (1, 0):
        ctx.tune(early_collapse=collapse)
        try:
            ck, cc, st = ctx.kmerize(d, K, native.KMERIZE_CANONICAL_ONLY)
        finally:
            ctx.tune(early_collapse=1)
        assert np.array_equal(ck.to_host(), ck_want) and np.array_equal(cc.to_host().astype(np.uint64), cc_want)
        assert st.n_unique == len(ck_want) and st.n_canonical == len(ck_want) and list(st.acgt) == want["acgt"]
        k, c = ctx.mirror_expand(ck, cc, K)
        assert np.array_equal(k.to_host(), wk) and np.array_equal(c.to_host(), wc)
    e = ctx.empty(0, np.uint64).view(0), ctx.empty(0, np.uint32).view(0)
    assert ctx.mirror_expand(e[0], e[1], K)[0].n == 0
    with pytest.raises(native.ZotkError):
        ctx.kmerize(d, K, native.KMERIZE_CANONICAL_ONLY | native.KMERIZE_BOTH)
    with pytest.raises(native.ZotkError):
        ctx.kmerize(d, K, native.KMERIZE_CANONICAL_ONLY | native.KMERIZE_SUBSAMPLE, p=0.5)
    if K & 1:
        # odd K: a canonical list and its mirror image share no key, and the union of the two is made without a tile waiting for
        # another (setops.hip, `disjoint`); a list that is NOT canonical (both strands in it) must be refused, not merged wrongly
        with pytest.raises(native.ZotkError):
            ctx.mirror_expand(ctx.upload(wk), ctx.upload(wc.astype(np.uint32)), K)
        k, c = ctx.mirror_expand(ck, cc, K)          # (and the context is fine afterwards)
        assert np.array_equal(k.to_host(), wk) and np.array_equal(c.to_host(), wc)


@pytest.mark.parametrize("case", ["u150", "u150_k13", "u100", "u40", "u255", "alt_149_151", "last_short", "n_at_separator", "one_read"])
def test_kmerize_record_aligned_tiles_and_fallbacks(ctx, case):
    """Pass 0 lays its tiles along the records when every record has the same length (checked on the device); any other
    stream takes tiles of positions.  Same answer either way, also for streams built to look uniform but are not."""
    rng = np.random.default_rng(len(case))

    def rnd(n, p_n=0.002):
        a = rng.choice(list("ACGT"), size=n)
        a[rng.random(n) < p_n] = "N"
        return "".join(a)
    K = 25
    if case == "u150":
        reads = [rnd(150) for _ in range(1333)]                    # not a multiple of the 64 records of a tile
    elif case == "u150_k13":
        K, reads = 13, [rnd(150) for _ in range(700)]              # 9 chunks cover 144 > 138 windows: the tail is masked
    elif case == "u100":
        reads = [rnd(100) for _ in range(2000)]                    # 5 chunks per record, 96 records per tile
    elif case == "u40":
        reads = [rnd(40) for _ in range(3000)]                     # one chunk per record: too many bytes per tile, falls back
    elif case == "u255":
        reads = [rnd(255) for _ in range(600)]
    elif case == "alt_149_151":
        reads = [rnd(149 if i % 2 == 0 else 151) for i in range(1500)]      # same total per pair, never uniform
    elif case == "last_short":
        reads = [rnd(150) for _ in range(900)] + [rnd(77)]
    elif case == "n_at_separator":
        reads = [rnd(150) for _ in range(500)]
        reads[3] = reads[3][:100] + "N" + reads[3][101:]           # fine: an N inside a record
        reads[7] = rnd(301)                                          # a long record whose middle byte sits on the separator grid
        reads[7] = reads[7][:150] + "N" + reads[7][151:]
    else:
        reads = [rnd(150)]
    want = zo.kmerize(K, reads)
    k, c, st = ctx.kmerize(ctx.upload_stream(stream_of(reads)), K)
    assert np.array_equal(k.to_host(), want["kmers"]) and np.array_equal(c.to_host(), want["counts"])
    assert list(st.acgt) == want["acgt"]


@pytest.mark.parametrize("shape", ["u150", "var_100_150", "u100_k31", "fasta_like_long", "lower_and_u", "poly_a", "u150_k9", "u150_k32"])
def test_kmerize_stream_ranges_pass(ctx, shape):
    """The first sort pass over static stream ranges (stream_pass.hip): one workgroup walks a range tile by tile, keeps what is
    left of a digit (fewer keys than a store unit) in LDS for the next tile, and writes whole units only.  With 3, 7 and the
    default number of ranges (a range of many tiles, of few, of at most one), both ways of sending a tile's units off (at once, in two bursts), with and without the look before
    the sort, the arrays must be the oracle's; and the look-back pipeline (stream_pass = 0) must still agree."""
    rng = np.random.default_rng(sum(map(ord, shape)))

    def rnd(n, p_n=0.001, alphabet="ACGT"):
        a = rng.choice(list(alphabet), size=n)
        a[rng.random(n) < p_n] = "N"
        return "".join(a)
    K = 25
    genome = rnd(30000, 0.0)

    def sampled(L):
        p = int(rng.integers(0, len(genome) - L))
        return genome[p:p + L]
    if shape == "u150":
        reads = [sampled(150) for _ in range(30000)]                        # ~470 record tiles, repeats its k-mers
    elif shape == "var_100_150":
        reads = [sampled(int(rng.integers(100, 151))) for _ in range(30000)]    # never uniform: tiles of positions
    elif shape == "u100_k31":
        K, reads = 31, [sampled(100) for _ in range(20000)]
    elif shape == "fasta_like_long":
        reads = [rnd(int(rng.integers(5000, 60000)), 0.0005) for _ in range(40)]   # records longer than a tile
    elif shape == "lower_and_u":
        reads = [rnd(150, 0.002, "ACGTacgtUu") for _ in range(20000)]
    elif shape == "poly_a":
        reads = ["A" * 150] * 6000 + [sampled(150) for _ in range(6000)] + ["T" * 150] * 3000      # one digit takes whole tiles
    elif shape == "u150_k9":
        K, reads = 9, [sampled(150) for _ in range(8000)]                    # 18 key bits: two passes of 9, nothing above bit 32
    else:
        K, reads = 32, [sampled(150) for _ in range(8000)]
    want = zo.kmerize(K, reads)
    d = ctx.upload_stream(stream_of(reads))
    try:
        for ranges in (3, 7, 0):
            for variant in (1, 3, 0):
                for collapse in ((1, 0) if variant == 1 else (1,)):
                    ctx.tune(stream_pass=variant, stream_ranges=ranges, early_collapse=collapse)
                    k, c, st = ctx.kmerize(d, K)
                    assert np.array_equal(k.to_host(), want["kmers"]), (shape, ranges, variant, collapse)
                    assert np.array_equal(c.to_host(), want["counts"]), (shape, ranges, variant, collapse)
                    assert list(st.acgt) == want["acgt"] and st.n_unique == len(want["kmers"]), (shape, ranges, variant, collapse)
        # both strands sorted literally: pass 0 emits x and rc(x) -- the pipeline's path, whatever the knob says
        ctx.tune(stream_pass=1, stream_ranges=0, early_collapse=1)
        k, c, st = ctx.kmerize(d, K, native.KMERIZE_BOTH)
        assert np.array_equal(k.to_host(), want["kmers"]) and np.array_equal(c.to_host(), want["counts"])
    finally:
        ctx.tune(stream_pass=1, stream_ranges=0, early_collapse=1)


@pytest.mark.parametrize("K", [4, 24, 32])
def test_kmerize_even_K_palindromes_vs_oracle(ctx, K):
    # even K: x == rc(x) exists; the mirrored path must count such a window twice, like two emissions
    reads = synth.read_strings(3, 0, 2500, 80, genome=0, n_thr=synth.frac32(0.01)) + ["ACGT" * 20, "AATT" * 16, "GC" * 40]
    want = zo.kmerize(K, reads)
    d = ctx.upload_stream(stream_of(reads))
    for flags in (native.KMERIZE_CANONICAL, native.KMERIZE_BOTH):
        k, c, st = ctx.kmerize(d, K, flags)
        assert np.array_equal(k.to_host(), want["kmers"]) and np.array_equal(c.to_host(), want["counts"])


@pytest.mark.parametrize("K", [1, 2, 13, 25, 31, 32])
def test_stream_checksum_is_an_independent_encoder(ctx, K):
    """zk_stream_checksum -- the checker behind every full-size run -- walks the stream byte by byte with the reference's
    rolling state (basics.kmersList) and shares no code with the product's tile encoder.  Against the oracle's k-mer lists:
    sums of 1, x and murmer(x) over both strands, and the acgt counts; reads shorter than K, N runs, lower case and U,
    pieces that start inside a read (128 bytes per thread)."""
    reads = synth.read_strings(5, 0, 1500, 131, genome=0, n_thr=synth.frac32(0.02))
    reads += ["", "A", "ACGT", "acgun" * 30, "N" * 200, "T" * 300, "ACGTTGCA" * 40, "GATTACA"]
    xs = np.concatenate([zo.kmers_list(K, r, True) for r in reads] + [np.zeros(0, dtype=np.uint64)])
    M = (1 << 64) - 1
    want = [len(xs), int(xs.sum(dtype=np.uint64)), sum(zo.murmer(int(x), 0) for x in xs) & M]
    want += [int(np.sum((xs & np.uint64(3)) == np.uint64(b))) for b in range(4)]
    d = ctx.upload_stream(stream_of(reads))
    assert [int(v) for v in ctx.stream_checksum(d, K)] + ctx.stream_acgt(d, K) == want


def test_kmerize_large_properties(ctx):
    """4.1 M reads x 150 bp: too big for the fixtures; checked through order-free checksums taken straight
    from the stream, sortedness, strand symmetry, and agreement of the sort strategies.  At this size (> 2^29 stream bytes) the
    default path is the block dedupe after two passes with 32-bit table entries, as on BASELINE config 2."""
    R, L, K = 4_100_000, 150, 25
    d = ctx.synth_reads(synth.DEFAULT_SEED, 0, R, L, genome=4_000_000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.0005))
    want = ctx.stream_checksum(d, K)
    k, c, st = ctx.kmerize(d, K, native.KMERIZE_CANONICAL, cap=3 * R * (L - K + 1) // 2)
    assert ctx.checksum(k, c) == want and st.n_instances == want[0]
    kh, ch = k.to_host(), c.to_host()
    assert np.all(kh[1:] > kh[:-1])
    assert ch.sum(dtype=np.uint64) == want[0]
    # strand symmetry: count(x) == count(rc x)
    sub = np.arange(0, len(kh), 997)
    rc = np.array([zo.rc(K, int(x)) for x in kh[sub]], dtype=np.uint64)
    j = np.searchsorted(kh, rc)
    assert np.array_equal(kh[j], rc) and np.array_equal(ch[j], ch[sub])
    k2, c2, st2 = ctx.kmerize(d, K, native.KMERIZE_BOTH, cap=len(kh) + 16)
    assert np.array_equal(k2.to_host(), kh) and np.array_equal(c2.to_host(), ch)
    assert list(st2.acgt) == list(st.acgt)
    try:
        for mode in (3, 2, 0):                   # tile-local ranking; run-length pass of its own; the plain full-width sort
            ctx.tune(early_collapse=mode)
            k4, c4, _ = ctx.kmerize(d, K, native.KMERIZE_CANONICAL, cap=len(kh) + 16)
            assert np.array_equal(k4.to_host(), kh) and np.array_equal(c4.to_host(), ch), mode
    finally:
        ctx.tune(early_collapse=1)
    # the same reads through the oracle on a 20 000-read prefix
    pre = ctx.synth_reads(synth.DEFAULT_SEED, 0, 20000, L, genome=4_000_000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.0005))
    reads = pre.to_host().tobytes().decode().split("\n")[:-1]
    w = zo.kmerize(K, reads)
    k3, c3, _ = ctx.kmerize(pre, K)
    assert np.array_equal(k3.to_host(), w["kmers"]) and np.array_equal(c3.to_host(), w["counts"])


def test_kmerize_large_without_repeats(ctx):
    """4.1 M random reads (no genome: hardly a k-mer occurs twice): the histogram kernel's look before the sort -- four whole
    blocks of the block dedupe set aside and counted -- must decline the top-bits-first plan, and the result must be that of the
    plain full-width sort and carry the stream's checksums."""
    R, L, K = 4_100_000, 150, 25
    d = ctx.synth_reads(synth.DEFAULT_SEED + 1, 0, R, L, genome=0, n_thr=synth.frac32(0.0005))
    want = ctx.stream_checksum(d, K)
    ctx.profile(True)
    k, c, st = ctx.kmerize(d, K, native.KMERIZE_CANONICAL, cap=2 * R * (L - K + 1) + 16)
    prof = ctx.profile_read()
    ctx.profile(False)
    assert ctx.checksum(k, c) == want and st.n_instances == want[0]
    assert prof["hist_stream"]["launches"] == 2 and prof["pass_stream"]["launches"] == 1, "declined before any pass, then planned again"
    kh = k.to_host()
    assert np.all(kh[1:] > kh[:-1])
    ctx.tune(early_collapse=0)
    try:
        k0, c0, _ = ctx.kmerize(d, K, native.KMERIZE_CANONICAL, cap=len(kh) + 16)
        assert np.array_equal(k0.to_host(), kh) and np.array_equal(c0.to_host(), c.to_host())
    finally:
        ctx.tune(early_collapse=1)


# ---- K5/K6 union-sum ----------------------------------------------------------------------------------------------

def test_merge_golden(ctx):
    parts = [G.load_case("g4_part%d" % i) for i in range(5)]
    dev = [(ctx.upload(p[1]), ctx.upload(p[2])) for p in parts]
    for n in (2, 3, 4, 5):
        info, km, ct, _, _ = G.load_case("g4_merge%d" % n)
        k, c, acgt = ctx.merge_n(dev[:n])
        assert np.array_equal(k.to_host(), km) and np.array_equal(c.to_host(), ct)
        assert {str(a): b for a, b in ctx.hist(c).items()} == info["meta"]["hist"]
        if n > 2:
            tot = float(sum(acgt))
            assert [v / tot for v in acgt] == info["meta"]["acgt"]     # merge.py:159,245-246
    # 2-way, both count widths
    _, km, ct, _, _ = G.load_case("g4_merge2")
    k, c = ctx.union_sum(dev[0][0], dev[0][1], dev[1][0], dev[1][1])
    assert np.array_equal(k.to_host(), km) and np.array_equal(c.to_host(), ct)
    a32 = (dev[0][0], ctx.upload(parts[0][2].astype(np.uint32)))
    b32 = (dev[1][0], ctx.upload(parts[1][2].astype(np.uint32)))
    k, c = ctx.union_sum(a32[0], a32[1], b32[0], b32[1])
    assert np.array_equal(k.to_host(), km) and np.array_equal(c.to_host().astype(np.uint64), ct)


@pytest.mark.parametrize("nx,ny", [(0, 0), (0, 10), (10, 0), (1, 1), (2047, 2049), (100000, 3), (300000, 250000)])
def test_union_sum_random_vs_oracle(ctx, nx, ny):
    rng = np.random.default_rng(nx * 7 + ny)
    pool = np.sort(rng.choice(np.arange(1 << 22, dtype=np.uint64), size=max(nx + ny, 1), replace=False))
    x = np.sort(rng.choice(pool, size=nx, replace=False)) if nx else np.empty(0, np.uint64)
    y = np.sort(rng.choice(pool, size=ny, replace=False)) if ny else np.empty(0, np.uint64)
    xc = rng.integers(1, 1000, size=nx, dtype=np.uint64)
    yc = rng.integers(1, 1000, size=ny, dtype=np.uint64)
    zs, zc = zo.union_sum(x, xc, y, yc)
    k, c, acgt = ctx.union_sum(ctx.upload(x), ctx.upload(xc), ctx.upload(y), ctx.upload(yc), want_acgt=True)
    assert np.array_equal(k.to_host(), zs) and np.array_equal(c.to_host(), zc)
    assert acgt == [int(zc[(zs & np.uint64(3)) == np.uint64(b)].sum()) for b in range(4)]
    # identical inputs: everything pairs up
    k, c = ctx.union_sum(ctx.upload(x), ctx.upload(xc), ctx.upload(x), ctx.upload(xc))
    assert np.array_equal(k.to_host(), x) and np.array_equal(c.to_host(), 2 * xc)


def test_merge_n_random_vs_oracle(ctx):
    """zk_merge_n: up to 16 lists per pass (kway.hip: sampled splitters, tiles merged in LDS) and the tree of 2-way passes (kway = 0)
    against the oracle's mergeNinto -- list counts around the fan-in (16, 17, 40: two levels), empty lists, lists that share most of
    their keys (long runs of equal keys across lists), a list far longer than the others, tiny inputs (fewer keys than one sample step)."""
    rng = np.random.default_rng(99)
    try:
        for k in (1, 2, 3, 7, 8, 13, 16, 17, 40):
            for shape in ("spread", "shared", "skewed", "tiny"):
                sets = []
                pool = np.arange(1 << 18, dtype=np.uint64) << np.uint64(30)
                if shape == "shared":
                    pool = pool[:60000]
                for s in range(k):
                    n = int(rng.integers(0, 40000))
                    if shape == "skewed":
                        n = 200000 if s == 1 else int(rng.integers(0, 300))
                    if shape == "tiny":
                        n = int(rng.integers(0, 40))
                    if s == 2:
                        n = 0
                    x = np.sort(rng.choice(pool, size=min(n, len(pool)), replace=False))
                    sets.append((x, rng.integers(1, 50, size=len(x), dtype=np.uint64)))
                zs, zc, acgt = zo.merge_n(25, sets)
                dev = [(ctx.upload(a), ctx.upload(b)) for a, b in sets]
                for kway in (2, 0):
                    ctx.tune(kway=kway)
                    gk, gc, gacgt = ctx.merge_n(dev)
                    assert np.array_equal(gk.to_host(), zs) and np.array_equal(gc.to_host(), zc) and gacgt == acgt, (k, shape, kway)
    finally:
        ctx.tune(kway=1)


# ---- K8/K9 dist --------------------------------------------------------------------------------------------------

def test_dist_golden(ctx):
    g = G.load_json("g5_dist")
    p0 = ctx.upload(G.load_case("g4_part0")[1])
    p1 = ctx.upload(G.load_case("g4_part1")[1])
    for k, t in g["split_part0_part1"].items():
        sh = 2 * (25 - int(k))
        x, y = ctx.project_dedupe(p0, sh), ctx.project_dedupe(p1, sh)
        assert (x.n, y.n) == (t["nx"], t["ny"])
        assert list(ctx.split(x, y)) == t["abc"]


def test_first_descent_and_measure_prep(ctx):
    """zk_first_descent: the first index whose k-mer is not above its predecessor (n if strictly ascending).  Measure.prep of
    zot dist (commands/dist.py:43-49) at the file's own K uses it to skip the identity copy; a set that repeats a k-mer, or a
    projection to a shorter K, still goes through zk_project_dedupe."""
    from zotmer_amd.library import engine
    rng = np.random.default_rng(5)
    k = np.unique(rng.integers(0, 1 << 50, size=300000, dtype=np.uint64))
    d = ctx.upload(k)
    assert ctx.first_descent(d) == d.n
    assert ctx.first_descent(ctx.upload(k[:1])) == 1 and ctx.first_descent(d.view(0)) == 0
    for at in (1, 77777, len(k) - 1):
        bad = k.copy()
        bad[at] = bad[at - 1]                               # equal neighbours are a descent too (strict)
        assert ctx.first_descent(ctx.upload(bad)) == at
    bad = k.copy()
    bad[1000], bad[200000] = bad[999] - np.uint64(1), np.uint64(0)
    assert ctx.first_descent(ctx.upload(bad)) == 1000
    assert engine.measure_prep(ctx, d, 0) is d             # nothing copied
    twice = np.sort(np.concatenate([k, k[:50]]))
    assert np.array_equal(engine.measure_prep(ctx, ctx.upload(twice), 0).to_host(), k)
    assert np.array_equal(engine.measure_prep(ctx, d, 14).to_host(), np.unique(k >> np.uint64(14)))


@pytest.mark.parametrize("nx,ny", [(0, 0), (0, 5), (5, 0), (1, 1), (4096, 4096), (100001, 77), (400000, 380000)])
def test_split_random_vs_oracle(ctx, nx, ny):
    rng = np.random.default_rng(nx + 3 * ny)
    pool = np.arange(1, 1 << 21, dtype=np.uint64) * np.uint64(0x1f3)
    x = np.sort(rng.choice(pool, size=nx, replace=False)) if nx else np.empty(0, np.uint64)
    y = np.sort(rng.choice(pool, size=ny, replace=False)) if ny else np.empty(0, np.uint64)
    assert ctx.split(ctx.upload(x), ctx.upload(y)) == zo.split(x, y)
    assert ctx.split(ctx.upload(x), ctx.upload(x)) == (nx, 0, 0)
    for sh in (0, 4, 20):
        assert np.array_equal(ctx.project_dedupe(ctx.upload(x), sh).to_host(), zo.project_dedupe(x, sh))


# ---- K10 trim -------------------------------------------------------------------------------------------------------

def test_trim_golden(ctx):
    _, km, ct, _, _ = G.load_case("g3_kmerize_genome")
    dk = ctx.upload(km)
    for name in ("g6_trim_c3", "g6_trim_c2_C9"):
        info, tk, tc, _, _ = G.load_case(name)
        for cdt in (np.uint64, np.uint32):
            k, c = ctx.trim(dk, ctx.upload(ct.astype(cdt)), info["c"], info["C"])
            assert np.array_equal(k.to_host(), tk) and np.array_equal(c.to_host().astype(np.uint64), tc)
    k, c = ctx.trim(dk, ctx.upload(ct), 10 ** 9)
    assert k.n == 0


# ---- f3: project / sample -----------------------------------------------------------------------------

@pytest.mark.parametrize("nr,ni", [(0, 0), (0, 1000), (1000, 0), (1, 1), (5000, 77), (77, 5000), (300_000, 1_000_000)])
def test_project_random_vs_oracle(ctx, nr, ni):
    rng = np.random.default_rng(7 * nr + ni)
    pool = np.arange(1, 1 << 21, dtype=np.uint64) * np.uint64(0x2b1)
    ref = np.sort(rng.choice(pool, size=nr, replace=False)) if nr else np.empty(0, np.uint64)
    x = np.sort(rng.choice(pool, size=ni, replace=False)) if ni else np.empty(0, np.uint64)
    c = rng.integers(1, 1 << 40, size=ni, dtype=np.uint64)
    k, kc = ctx.project(ctx.upload(ref), ctx.upload(x), ctx.upload(c))
    ek, ec = zo.project(ref, x, c)
    assert np.array_equal(k.to_host(), ek) and np.array_equal(kc.to_host(), ec)
    k, kc = ctx.project(ctx.upload(x), ctx.upload(x), ctx.upload(c))          # a set projected on itself is itself
    assert np.array_equal(k.to_host(), x) and np.array_equal(kc.to_host(), c)


def test_project_and_sample_golden(ctx):
    _, k0, c0, _, _ = G.load_case("g4_part0")
    _, k1, _, _, _ = G.load_case("g4_part1")
    _, ek, ec, _, _ = G.load_case("f3_project_part0_on_part1")
    k, c = ctx.project(ctx.upload(k1), ctx.upload(k0), ctx.upload(c0))
    assert np.array_equal(k.to_host(), ek) and np.array_equal(c.to_host(), ec)
    for name in ("f3_sample_D_S5_P0.3", "f3_sample_defaults"):
        info, ek, ec, _, _ = G.load_case(name)
        k, c = ctx.sample(ctx.upload(k0), ctx.upload(c0), info["S"], info["P"])
        assert np.array_equal(k.to_host(), ek) and np.array_equal(c.to_host(), ec)


@pytest.mark.parametrize("n,p,seed", [(0, 0.5, 1), (1000, 0.0, 1), (1000, 1.0, 1), (1000, 1.5, 9), (2_000_000, 0.01, 0),
                                      (2_000_000, 0.37, 2 ** 63 + 11)])
def test_sample_random_vs_oracle(ctx, n, p, seed):
    rng = np.random.default_rng(n + seed % 1000)
    x = np.sort(rng.integers(0, 1 << 50, size=n, dtype=np.uint64))
    c = rng.integers(1, 1 << 33, size=n, dtype=np.uint64)
    k, kc = ctx.sample(ctx.upload(x), ctx.upload(c), seed, p)
    ek, ec = zo.sample_d(p, seed, x, c)
    assert np.array_equal(k.to_host(), ek) and np.array_equal(kc.to_host(), ec)
    if p >= 1.0 and n:
        # u == 1.0 exactly only when all 40 hash bits are set; with p > 1 everything stays
        assert k.n == n if p > 1.0 else k.n >= n - 1
