"""K11 / K12: the codec64 + delta codec on the device, byte-exact against the reference's words and
raw streams (tests/golden) and against the oracle on large random inputs that cross many tiles."""
import numpy as np
import pytest

from oracle import zkoracle as zo
from tests import _golden as G
from zotmer_amd import native

pytestmark = pytest.mark.gpu
P = G.load_json("primitives")


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


def test_reference_words(ctx):
    for c in P["codec64"]:
        v = np.array(c["values"], dtype=np.uint64)
        if len(v) == 0:
            continue
        w = ctx.codec_encode(ctx.upload(v), False).to_host()
        assert [int(x) for x in w] == c["words"]
        back = ctx.codec_decode(ctx.upload(np.array(c["words"], dtype=np.uint64)), False).to_host()
        assert [int(x) for x in back] == c["values"]


def test_errors(ctx):
    for c in P["codec64_errors"]:
        with pytest.raises(native.ZotkError) as e:
            ctx.codec_encode(ctx.upload(np.array(c["values"], dtype=np.uint64)), False)
        assert e.value.code == native.ZK_ERANGE
    for c in P["codec64_decode_tags"]:
        w = ctx.upload(np.array([c["word"]], dtype=np.uint64))
        if "error" in c:
            with pytest.raises(native.ZotkError) as e:
                ctx.codec_decode(w, False)
            assert e.value.code == native.ZK_ERANGE
        else:
            assert [int(x) for x in ctx.codec_decode(w, False).to_host()] == c["out"]
    # k-mer deltas >= 2^60 (the K=31 poly-T file the reference cannot write)
    with pytest.raises(native.ZotkError):
        ctx.codec_encode(ctx.upload(np.array([0, (1 << 62) - 1], dtype=np.uint64)), True)


@pytest.mark.parametrize("name", ["g2_kmerize_uniformN", "g3_kmerize_genome", "g3_kmerize_genome_k12", "g8_kmerize_k31",
                                  "g4_merge5", "g6_trim_c3", "g9_edge_fastq"])
def test_raw_streams_byte_exact(ctx, name):
    info, km, ct, raw_k, raw_c = G.load_case(name)
    assert ctx.codec_encode(ctx.upload(km), True).to_host().astype("<u8").tobytes() == raw_k
    assert ctx.codec_encode(ctx.upload(ct), False).to_host().astype("<u8").tobytes() == raw_c
    assert np.array_equal(ctx.codec_decode(ctx.upload(np.frombuffer(raw_k, dtype="<u8")), True, len(km)).to_host(), km)
    assert np.array_equal(ctx.codec_decode(ctx.upload(np.frombuffer(raw_c, dtype="<u8")), False).to_host(), ct)


@pytest.mark.parametrize("n", [1, 31, 32, 33, 8191, 8192, 8193, 300001, 3_000_000])
def test_random_many_tiles_vs_oracle(ctx, n):
    rng = np.random.default_rng(n)
    widths = rng.choice([0, 1, 3, 9, 10, 11, 12, 13, 15, 16, 20, 21, 30, 31, 59, 60], size=n)
    v = np.array([int(rng.integers(0, 1 << 62)) & ((1 << int(w)) - 1) for w in widths[:min(n, 50000)]], dtype=np.uint64)
    if n > 50000:
        v = np.concatenate([v, rng.integers(0, 1 << 20, size=n - 50000, dtype=np.uint64) >> rng.integers(0, 20, size=n - 50000).astype(np.uint64)])
    want = zo.codec64_encode(v)
    got = ctx.codec_encode(ctx.upload(v), False)
    assert np.array_equal(got.to_host(), want)
    assert np.array_equal(ctx.codec_decode(got, False).to_host(), v)
    # k-mers: ascending 50-bit (and 64-bit cumulative) values through the delta form
    k = np.sort(rng.integers(0, 1 << 50, size=n, dtype=np.uint64))
    wk = ctx.codec_encode(ctx.upload(k), True)
    assert np.array_equal(wk.to_host(), zo.codec64_encode(zo.delta(k)))
    assert np.array_equal(ctx.codec_decode(wk, True, n).to_host(), k)
    big = np.cumsum(rng.integers(0, 1 << 44, size=min(n, 1 << 19), dtype=np.uint64), dtype=np.uint64)   # reaches beyond 2^57
    wb = ctx.codec_encode(ctx.upload(big), True)
    assert np.array_equal(ctx.codec_decode(wb, True, len(big)).to_host(), big)


def test_encode_u32_counts_and_the_grown_word_buffer(ctx):
    """32-bit counts are encoded without being widened (zk_codec64_encode_u32_dev): the same words as the 64-bit encoder gives
    for the same values, and as the host codec (library/codec64.py:82-120).  Values of more than 30 bits take a word each: more
    words than the half-a-word-per-value buffer the wrapper starts with, so the grown buffer is exercised too."""
    rng = np.random.default_rng(17)
    for n, hi in ((0, 4), (1, 4), (100003, 4), (300000, 12), (200000, 32), (70000, 31)):
        v = (rng.integers(0, 1 << 62, size=n, dtype=np.uint64) >> rng.integers(64 - hi, 64, size=n, dtype=np.uint64).astype(np.uint64)).astype(np.uint32)
        if n > 5000 and hi >= 31:
            v[1000:60000] |= np.uint32(1 << 30)            # a long stretch of one-value words
        want = zo.codec64_encode(v.astype(np.uint64))
        got32 = ctx.codec_encode(ctx.upload(v), False).to_host() if n else np.empty(0, np.uint64)
        got64 = ctx.codec_encode(ctx.upload(v.astype(np.uint64)), False).to_host() if n else np.empty(0, np.uint64)
        assert np.array_equal(got32, want) and np.array_equal(got64, want), (n, hi)
