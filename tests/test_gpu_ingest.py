"""
f2 ingest on the GPU box (csrc/ingest.hip + library/engine.py count_fastq_file): FASTQ files -- plain, gzip, multi-member gzip --
read ahead by the native reader, cut into batches on the device and parsed there, against the oracle on the same reads;
tiny batch sizes force many batches, carried line tails and ring wrap-arounds.  Plus the device <-> file member path.
"""
import gzip
import os

import numpy as np
import pytest

from oracle import zkoracle as zo
from zotmer_amd import native, synth
from zotmer_amd.library import engine

pytestmark = pytest.mark.gpu
K = 25


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    engine.release_table_memory(c)
    c.close()


def _reads(n, first=0):
    return synth.read_strings(synth.DEFAULT_SEED, first, n, 150, genome=30000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.001))


def _fastq(reads, first=0):
    return "".join("@r%d\n%s\n+\n%s\n" % (first + i, s, "I" * len(s)) for i, s in enumerate(reads)).encode()


def _check(ctx, path, reads, batch):
    want = zo.kmerize(K, reads)
    t = engine.KmerTable(ctx, K)
    recs = engine.count_fastq_file(ctx, t, str(path), batch)
    k, c, h = t.result()
    assert recs == len(reads)
    assert np.array_equal(k, want["kmers"]) and np.array_equal(c, want["counts"]) and t.acgt == want["acgt"]
    hv, hf = zo.hist(want["counts"])
    assert h == {int(a): int(b) for a, b in zip(hv, hf)}


@pytest.mark.parametrize("batch", [4096, 100_000, 1 << 20, 64 << 20])
def test_plain_fastq_batches(ctx, tmp_path, batch):
    reads = _reads(3000)
    p = tmp_path / "a.fastq"
    p.write_bytes(_fastq(reads))
    _check(ctx, p, reads, batch)


def test_gzip_and_multi_member_gzip(ctx, tmp_path):
    reads = _reads(4000)
    text = _fastq(reads)
    p = tmp_path / "a.fastq.gz"
    p.write_bytes(gzip.compress(text, 6))
    _check(ctx, p, reads, 200_000)
    # concatenated members (bgzip / `cat a.gz b.gz`), the cut in the middle of a line
    cut = len(text) // 3 + 17
    q = tmp_path / "b.fastq.gz"
    q.write_bytes(gzip.compress(text[:cut], 1) + gzip.compress(text[cut:2 * cut], 9) + gzip.compress(text[2 * cut:], 5))
    _check(ctx, q, reads, 150_000)


def test_edge_inputs(ctx, tmp_path):
    reads = _reads(50)
    text = _fastq(reads)
    cases = {
        "no_final_newline": (text[:-1], reads),
        "incomplete_record_1_line": (text + b"@extra\n", reads),
        "incomplete_record_2_lines": (text + b"@extra\n" + reads[0].encode() + b"\n", reads),          # its sequence line must NOT count
        "incomplete_record_3_lines_no_nl": (text + b"@extra\n" + reads[1].encode() + b"\n+", reads),
        "empty": (b"", []),
        "one_record": (_fastq(reads[:1]), reads[:1]),
        "lower_case_and_N": (_fastq([r.lower().replace("a", "N", 1) for r in reads]), [r.lower().replace("a", "N", 1) for r in reads]),
    }
    for name, (data, rs) in cases.items():
        p = tmp_path / (name + ".fastq")
        p.write_bytes(data)
        for batch in (700, 1 << 16):
            if rs:
                _check(ctx, p, rs, batch)
            else:
                t = engine.KmerTable(ctx, K)
                assert engine.count_fastq_file(ctx, t, str(p), batch) == 0 and t.result()[0].size == 0


def test_line_longer_than_batch_is_reported(ctx, tmp_path):
    p = tmp_path / "long.fastq"
    p.write_bytes(b"@r\n" + b"A" * 5000 + b"\n+\n" + b"I" * 5000 + b"\n")
    t = engine.KmerTable(ctx, K)
    with pytest.raises(IOError):
        engine.count_fastq_file(ctx, t, str(p), 1000)


def test_missing_file(ctx):
    with pytest.raises(IOError):
        ctx.source_open("/nonexistent/zot.fastq")


@pytest.mark.parametrize("n", [0, 1, 12345, (32 << 20) // 8, (100 << 20) // 8 + 3])
def test_device_file_round_trip(ctx, tmp_path, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    p = tmp_path / "m.bin"
    with open(p, "wb") as f:
        f.write(b"HEAD")
        f.flush()
        ctx.device_to_file(ctx.upload(a), f.fileno(), 4)
    raw = p.read_bytes()
    assert raw[:4] == b"HEAD" and raw[4:] == a.tobytes()
    with open(p, "rb") as f:
        back = ctx.file_to_device(f.fileno(), 4, 8 * n, np.uint64)
    assert np.array_equal(back.to_host(), a)


def test_table_slab_survives_growth_and_estimates(ctx):
    """batches of very different yield: the per-byte estimate from a poor batch is too small for a rich one (ZK_ENOSPC retry),
    and the slab grows while tables are live"""
    engine.release_table_memory(ctx)
    rich = _reads(3000, first=5000)
    poor = ["N" * 150] * 2000 + _reads(20)
    t = engine.KmerTable(ctx, K)
    for rs in (poor, rich, poor, rich[:700], _reads(900, first=20000)):
        t.add_stream(np.frombuffer(("".join(r + "\n" for r in rs)).encode(), dtype=np.uint8))
    want = zo.kmerize(K, poor + rich + poor + rich[:700] + _reads(900, first=20000))
    k, c, _ = t.result()
    assert np.array_equal(k, want["kmers"]) and np.array_equal(c, want["counts"]) and t.acgt == want["acgt"]
