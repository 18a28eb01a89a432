"""
The BASELINE configurations at their FULL size, through the C-ABI.  No oracle can run 12.6 G k-mer instances, so the checks are
the size-independent properties of the domain: order-free checksums of the result against the same sums
taken straight from the input by an independent encoder, strict ascent of the k-mers (the sorted-set format's invariant,
library/files.py:54-110), strand symmetry (count(x) == count(rc x): reads(..., both=True), library/reads.py:113-114), and for
`zot dist` an independent path (sort + run-length count of the concatenation).  A few seconds of GPU time each.
"""
import numpy as np
import pytest

from zotmer_amd import native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


def ctx_tag_words(ctx):
    """the library's default for ZK_TUNE_TAG_WORDS (1: the tag pass ranks by ballots, 2: by LDS adds inside a bucket); the other one is
    the variant the full-size test also runs"""
    return native.DEFAULT_TAG_WORDS


def revcomp(K, x):
    r = 0
    for _ in range(K):
        r = (r << 2) | (3 - (x & 3))
        x >>= 2
    return r


def test_config2_full_size(ctx):
    """BASELINE config 2: zot kmerize k=25 on 50 M x 150 bp genome-sampled reads (commands/kmerize.py:450-562)."""
    cfg = synth.CONFIGS["config2"]
    R, L, K = cfg["reads"], cfg["L"], cfg["K"]
    free, _ = ctx.mem_info()
    if free < 200 << 30:
        pytest.skip("needs ~170 GB of device memory")
    d = ctx.synth_reads(synth.DEFAULT_SEED, 0, R, L, genome=cfg["genome"], sub_thr=synth.frac32(cfg["sub"]), n_thr=synth.frac32(cfg["n"]))
    cap = int(2 * (cfg["genome"] + R * L * cfg["sub"] * 22) * 1.25) + (1 << 20)
    out = (ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32))
    k, c, st = ctx.kmerize(d, K, out=out)
    want = ctx.stream_checksum(d, K)
    assert ctx.checksum(k, c) == want                                   # sum 1, sum x, sum murmer(x) over every instance
    assert st.n_instances == want[0] and st.n_unique == k.n
    assert list(st.acgt) == list(ctx.stream_acgt(d, K)) and sum(st.acgt) == st.n_instances
    assert ctx.first_descent(k) == k.n                                  # strictly ascending: sorted, and no k-mer in two entries
    h = ctx.hist(c)
    assert sum(h.values()) == k.n and sum(v * n for v, n in h.items()) == st.n_instances
    # strand symmetry on a sample: rc(x) is in the table with the same count (K odd: never x itself)
    rng = np.random.default_rng(2)
    idx = np.sort(rng.integers(0, k.n, size=4096))
    kh, ch = k.to_host(), c.to_host()
    xs, cs = kh[idx], ch[idx]
    q = np.array([revcomp(K, int(x)) for x in xs], dtype=np.uint64)
    pos = ctx.lower_bound(k, q).astype(np.int64)
    assert np.array_equal(kh[pos], q) and np.array_equal(ch[pos], cs)
    # the same result from the look-back pipeline's first pass (the two first passes share no kernel)
    n_first = k.n
    del kh, ch
    try:
        ctx.tune(stream_pass=0)
        k0, c0, _ = ctx.kmerize(d, K, out=out)          # (into the same arrays: every check of the first result is done)
        assert k0.n == n_first and ctx.checksum(k0, c0) == want and ctx.first_descent(k0) == k0.n
    finally:
        ctx.tune(stream_pass=1)
    # ... from the second pass over static segments (tag_pass.hip; pass 0 then writes two arrays), and from the block dedupe with
    # one workgroup per CU (round 3's kernel; the default, dedupe2_kernel, made the first result), from the tag pass that ranks every
    # tile by ballots, and from that pass on 8 K-key tiles (the default: 16 K)
    for knob in (dict(tag_pass=1), dict(dedupe_variant=-1), dict(tag_words=3 - ctx_tag_words(ctx)), dict(wide_tiles=0)):
        try:
            ctx.tune(**knob)
            k0, c0, _ = ctx.kmerize(d, K, out=out)
            assert k0.n == n_first and ctx.checksum(k0, c0) == want and ctx.first_descent(k0) == k0.n, knob
        finally:
            ctx.tune(tag_pass=0, dedupe_variant=0, tag_words=ctx_tag_words(ctx), wide_tiles=1)
    del k0, c0, k, c, out, d
    ctx.release_workspace()


def test_config4_share_full_size(ctx):
    """One GPU's share of BASELINE config 4 (commands/merge.py:127-163): 8 sets of 50 M (k-mer, 64-bit count) pairs, union-summed in
    one pass (kway.hip) and by the tree of 2-way passes: the same table, its checksum of checksums the sum of the inputs', strictly
    ascending."""
    sets, sums, total = [], [0, 0, 0], 0
    for s in range(8):
        a = synth.config4_set_args(s, 1.0)
        k, c = ctx.synth_set(a["seed"], a["first"], a["count"], a["key_bits"], mul=a["mul"], add=a["add"], mod=a["mod"])
        sets.append((k, c))
        for i, v in enumerate(ctx.checksum_counts(k, c)):
            sums[i] = (sums[i] + v) & ((1 << 64) - 1)
        total += k.n
    out = (ctx.empty(total, np.uint64), ctx.empty(total, np.uint64))
    got = {}
    try:
        for kway in (1, 0):
            ctx.tune(kway=kway)
            mk, mc, acgt = ctx.merge_n(sets, out=out)
            assert list(ctx.checksum_counts(mk, mc)) == sums and ctx.first_descent(mk) == mk.n and sum(acgt) == sums[0], kway
            got[kway] = (mk.n, ctx.checksum_counts(mk, mc), list(acgt))
    finally:
        ctx.tune(kway=1)
    assert got[1] == got[0]
    del sets, out
    ctx.release_workspace()


def test_config3_full_size(ctx):
    """BASELINE config 3: zot dist on two sorted sets of 100 M 50-bit k-mers (commands/dist.py:94-168, library/dist.py:241-265):
    (a, b, c) of the merge-path split against the sort-and-count path and the generator's construction."""
    c3 = synth.CONFIG3
    n = c3["n"]
    ka, _ = ctx.synth_set(c3["seed"], 0, n, c3["key_bits"], counts=False)
    kb, _ = ctx.synth_set(c3["seed"], n // 2, n, c3["key_bits"], counts=False)
    assert ctx.first_descent(ka) == ka.n and ctx.first_descent(kb) == kb.n
    a, b, c = ctx.split(ka, kb)
    # Measure.prep at K == fK is the identity (commands/dist.py:43-49): the command's path gives the same triple
    pa, pb = ctx.project_dedupe(ka, 0), ctx.project_dedupe(kb, 0)
    assert pa.n == ka.n and pb.n == kb.n and ctx.split(pa, pb) == (a, b, c)
    del pa, pb
    cat = ctx.empty(ka.n + kb.n, np.uint64)
    ctx._check(ctx.lib.zk_copy(ctx.h, cat.ptr, ka.ptr, ka.nbytes))
    ctx._check(ctx.lib.zk_copy(ctx.h, cat.ptr + ka.nbytes, kb.ptr, kb.nbytes))
    ctx.sync()
    u, cnt = ctx.sort_count(cat, c3["key_bits"])
    h = ctx.hist(cnt)
    assert set(h) <= {1, 2} and h.get(2, 0) == a and u.n == a + b + c
    assert a + b == ka.n and a + c == kb.n
    ctx.release_workspace()
