"""
CPU tests of the host half of the drop-in (no GPU needed): native codec64 + delta against the
reference's words and raw streams (tests/golden), the container layout, the text parsers, the usage
parser and the distance formulas.  The reference's own two tests (tests/test_files.py:8-48) are
restated at the end.
"""
import ctypes as C
import fnmatch
import gzip
import json
import os
import random
import struct

import numpy as np
import pytest

from tests import _golden as G
from zotmer_amd import native
from zotmer_amd.library import measures, seqio, vectors
from zotmer_amd.library.container import Container, KmerSet
from zotmer_amd.library.usage import Spec, UsageError

P = G.load_json("primitives")
CASES = ["g2_kmerize_uniformN", "g3_kmerize_genome", "g3_kmerize_genome_k24", "g3_kmerize_genome_k12", "g8_kmerize_k31",
         "g9_edge_fastq", "g9_edge_fasta", "g9_two_files", "g10_kmerize_D0.8_S3", "g10_kmerize_capture",
         "g4_part0", "g4_merge2", "g4_merge3", "g4_merge5", "g6_trim_c3", "g6_trim_c2_C9"]


def test_codec64_words_match_reference():
    for c in P["codec64"]:
        w = np.frombuffer(vectors.encode_counts(np.array(c["values"], dtype=np.uint64)), dtype="<u8")
        assert [int(x) for x in w] == c["words"]
        assert [int(x) for x in vectors.decode_counts(w.tobytes())] == c["values"]
    for c in P["codec64_errors"]:
        with pytest.raises(vectors.CodecError):
            vectors.encode_counts(np.array(c["values"], dtype=np.uint64))
    for c in P["codec64_decode_tags"]:
        data = struct.pack("<Q", c["word"])
        if "error" in c:
            with pytest.raises(vectors.CodecError):
                vectors.decode_counts(data)
        else:
            assert [int(x) for x in vectors.decode_counts(data)] == c["out"]


@pytest.mark.parametrize("name", CASES)
def test_raw_streams_byte_exact(name):
    info, km, ct, raw_k, raw_c = G.load_case(name)
    assert vectors.encode_kmers(km) == raw_k
    assert vectors.encode_counts(ct) == raw_c
    assert np.array_equal(vectors.decode_kmers(raw_k), km)
    assert np.array_equal(vectors.decode_counts(raw_c), ct)


def test_k31_delta_overflow_is_an_error():
    # 'T'*31 -> k-mers {0, 2**62 - 1}: the reference dies writing it (tests/golden/g8_k31_delta_overflow.json)
    with pytest.raises(vectors.CodecError):
        vectors.encode_kmers(np.array([0, (1 << 62) - 1], dtype=np.uint64))


def test_container_layout(tmp_path):
    info, km, ct, raw_k, raw_c = G.load_case("g3_kmerize_genome")
    # a file assembled exactly as the reference lays it out (TOC from the golden run) reads back
    p = tmp_path / "ref.k25"
    meta = json.dumps(info["meta"]).encode()
    toc = json.dumps({"kmers": [[0, len(raw_k)]], "counts": [[len(raw_k), len(raw_c)]],
                      "__meta__": [[len(raw_k) + len(raw_c), len(meta)]]}).encode()
    p.write_bytes(raw_k + raw_c + meta + toc + struct.pack("<Q", len(toc)))
    assert info["toc"]["kmers"] == [[0, len(raw_k)]] and info["toc"]["counts"] == [[len(raw_k), len(raw_c)]]
    with KmerSet(str(p), "r") as z:
        assert z.meta == info["meta"]
        k, c = vectors.read_kmers_and_counts(z)
        assert np.array_equal(k, km) and np.array_equal(c, ct)
        assert dict(z.names())["kmers"] == len(raw_k)
    # and what this build writes has the same members at the same offsets
    q = tmp_path / "mine.k25"
    with KmerSet(str(q), "w") as z:
        vectors.write_kmers_and_counts(z, km, ct)
        z.meta.update(info["meta"])
    blob = q.read_bytes()
    (n,) = struct.unpack("<Q", blob[-8:])
    toc2 = json.loads(blob[-8 - n:-8])
    assert toc2["kmers"] == info["toc"]["kmers"] and toc2["counts"] == info["toc"]["counts"]
    assert toc2["__meta__"][0][0] == info["toc"]["__meta__"][0][0]
    assert blob[:len(raw_k)] == raw_k and blob[len(raw_k):len(raw_k) + len(raw_c)] == raw_c
    with KmerSet(str(q), "r") as z:
        assert z.meta == info["meta"]


def test_container_versions_and_stream(tmp_path):
    p = str(tmp_path / "c")
    with Container(p, "w") as z:
        z.add("a", b"one")
        with z.add_stream("b") as f:
            f.write(b"12")
            f.write(b"345")
        z.add("a", b"three")
    with Container(p, "r") as z:
        assert z.read("a") == b"three" and z.read("b") == b"12345"       # last version wins (casket.py:185)
        assert z.names() == [("a", 5), ("b", 5)]
        with pytest.raises(KeyError):
            z.read("zz")


def _streams(paths, **kw):
    out, recs = [], 0
    for s, r in seqio.base_stream_batches(paths, **kw):
        out.append(bytes(s))
        recs += r
    return b"".join(out), recs


def test_parsers_match_reference_readers(tmp_path):
    fq = G.load_json("g9_edge_fastq")["fastq"]
    fa = G.load_json("g9_edge_fasta")["fasta"]
    pq, pa = tmp_path / "x.fastq", tmp_path / "x.fa"
    pq.write_text(fq)
    pa.write_text(fa)
    s, r = _streams([str(pq)])
    assert s == "".join(x + "\n" for x in G.fastq_seqs(fq)).encode() and r == 6
    s, r = _streams([str(pa)])
    assert s == "".join(x + "\n" for x in G.fasta_seqs(fa)).encode() and r == 3
    # compressed input and tiny chunks (records split across chunk boundaries)
    gz = tmp_path / "y.fastq.gz"
    big = G.synth_fastq(G.load_json("g3_kmerize_genome"))
    with gzip.open(str(gz), "wb") as f:
        f.write(big.encode())
    want = "".join(x + "\n" for x in G.fastq_seqs(big)).encode()
    for chunk in (1 << 20, 777, 4096):
        s, r = _streams([str(gz)], chunk_bytes=chunk, batch_bytes=5000)
        assert s == want and r == 1500
    # file type by suffix (reads.py:24-33)
    assert seqio.is_fasta("a.fa") and seqio.is_fasta("a.fasta.gz") and seqio.is_fasta("b.fna.bz2") and seqio.is_fasta("c.fas")
    assert not seqio.is_fasta("a.fastq") and not seqio.is_fasta("a.fq.gz") and not seqio.is_fasta("fa")
    # text before the first FASTA header is ignored, CRLF line ends are stripped
    pa2 = tmp_path / "z.fa"
    pa2.write_bytes(b"junk\r\n>r1\r\nACGT\r\nAC\r\n\r\n>r2\r\n>r3\r\nGG")
    s, r = _streams([str(pa2)])
    assert s == b"ACGTAC\n\nGG\n" and r == 3


def test_usage_parser():
    sp = Spec(options={"-m": True, "-v": False, "-M": "list"}, positionals=["<k>", "<out>"], rest="<in>")
    o = sp.parse(["-v", "-m", "5", "-Mx", "-M", "y", "25", "o", "a", "b"], "usage")
    assert o == {"-m": "5", "-v": True, "-M": ["x", "y"], "<k>": "25", "<out>": "o", "<in>": ["a", "b"]}
    with pytest.raises(UsageError):
        sp.parse(["25", "o"], "usage")
    with pytest.raises(UsageError):
        sp.parse(["-q", "25", "o", "a"], "usage")


def test_measures_match_reference_output():
    g = G.load_json("g5_dist")
    listing = "\n".join(m + "\t" + measures.MEASURES[m][0] for m in sorted(measures.MEASURES)) + "\n"
    assert listing == g["list"]["stdout"]
    t = g["split_part0_part1"]["25"]["abc"]
    row = g["qual_k25"]["stdout"].split("\n")[1].split("\t")[2:]
    ms = sorted(m for m in measures.MEASURES if fnmatch.fnmatch(m, "*.qual"))
    assert ["%g" % measures.MEASURES[m][2](*t) for m in ms] == row


def test_reference_test_files_roundtrips(tmp_path):
    # tests/test_files.py:8-26 (test_rwVector): 65 536 small ints through writeVector / readVectorList
    random.seed(17)
    xs = np.array([int(0.5 + 10 * random.expovariate(0.5)) for _ in range(65536)], dtype=np.uint64)
    p = str(tmp_path / "v")
    with Container(p, "w") as z:
        z.add("quux", vectors.encode_counts(xs))
    with Container(p, "r") as z:
        assert np.array_equal(vectors.decode_counts(z.read("quux")), xs)
    # tests/test_files.py:28-48 (test_kmersList): 65 536 sorted random 50-bit k-mers through the delta form
    ks = np.sort(np.array([random.randint(0, (1 << 50) - 1) for _ in range(65536)], dtype=np.uint64))
    with Container(p, "w") as z:
        z.add("kmers", vectors.encode_kmers(ks))
    with Container(p, "r") as z:
        assert np.array_equal(vectors.decode_kmers(z.read("kmers")), ks)


def test_jstats_reproduces_the_reference_lines():
    """library/jstats.py (product host code) against the lines the reference printed (tests/golden/f3_jaccard.json)."""
    import json
    import os
    from zotmer_amd.library import jstats
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "f3_jaccard.json")))
    for key, p in (("jaccard_default", None), ("jaccard_all", None), ("jaccard_p0.5", 0.5), ("jaccard_fasta_all", None)):
        for line in g[key]["stdout"].strip().split("\n"):
            f = line.split("\t")
            if len(f) < 7:
                continue
            assert jstats.jaccard_fields(int(f[2]), int(f[3]), int(f[4]), p) == "\t".join(f[2:])
    assert jstats.log_choose(10, 0) == 0 and jstats.log_choose(10, 10) == 0
    import math
    assert abs(jstats.log_choose(40, 20) - math.log(math.comb(40, 20))) < 1e-6
    assert abs(jstats.quant_beta(0.5, 50, 50) - 0.5) < 1e-3


def test_fasta_records(tmp_path):
    from zotmer_amd.library import seqio
    p = tmp_path / "x.fa"
    p.write_text("stray\n>a one  \nACG\n TTA \n\n>b\n>c\nGG")
    assert list(seqio.fasta_records(str(p))) == [("a one", b"ACGTTA"), ("b", b""), ("c", b"GG")]
