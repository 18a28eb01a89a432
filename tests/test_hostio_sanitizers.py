"""
The product's host code -- zotmer_amd/csrc/hostio.cpp: codec64 + delta (zotmer/library/codec64.py:42-151,
library/files.py:85-110) and the FASTQ / FASTA chunk parsers (library/file.py:19-52) -- built with AddressSanitizer and
UndefinedBehaviorSanitizer and run on the CPU against the golden streams.  The driver
(tests/san/hostio_san_driver.cpp) hands the library heap blocks of exactly the advertised size, so an access one byte past a
buffer ends the run.  GPU sanitizers are not available on this pool; the kernels are covered by the parity tests.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "san", "hostio_san_driver")
CASES = ["g2_kmerize_uniformN", "g3_kmerize_genome_k24", "g8_kmerize_k31", "g9_edge_fastq", "g4_merge5", "g6_trim_c3"]


@pytest.fixture(scope="module")
def driver():
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("g++ / make not available")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "zotmer_amd", "csrc"), "hostio_san"], stdout=subprocess.DEVNULL)
    return DRIVER


def run(driver, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([driver] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, "sanitizer or driver failure:\n" + p.stdout + p.stderr[-4000:]
    return p.stdout


@pytest.mark.parametrize("name", CASES)
def test_golden_streams_under_sanitizers(driver, tmp_path, name):
    info, km, ct, raw_k, raw_c = G.load_case(name)
    for what, vals, raw, delta in (("k", km, raw_k, 1), ("c", ct, raw_c, 0)):
        v, w, o = tmp_path / (what + ".val"), tmp_path / (what + ".words"), tmp_path / (what + ".out")
        v.write_bytes(np.asarray(vals, dtype="<u8").tobytes())
        assert run(driver, "enc", delta, v, o).startswith("rc 0")
        assert o.read_bytes() == raw                              # the reference's bytes
        w.write_bytes(raw)
        assert run(driver, "dec", delta, w, o).startswith("rc 0")
        assert np.array_equal(np.frombuffer(o.read_bytes(), dtype="<u8"), np.asarray(vals, dtype=np.uint64))


def test_codec_edges_under_sanitizers(driver, tmp_path):
    P = G.load_json("primitives")
    v, o = tmp_path / "v", tmp_path / "o"
    for c in P["codec64"]:
        v.write_bytes(np.array(c["values"], dtype="<u8").tobytes())
        assert run(driver, "enc", 0, v, o).startswith("rc 0")
        assert [int(x) for x in np.frombuffer(o.read_bytes(), dtype="<u8")] == c["words"]
    for c in P["codec64_errors"]:
        v.write_bytes(np.array(c["values"], dtype="<u8").tobytes())
        assert run(driver, "enc", 0, v, o).startswith("rc -")    # refused, as the reference raises
    for c in P["codec64_decode_tags"]:
        v.write_bytes(np.array([c["word"]], dtype="<u8").tobytes())
        out = run(driver, "dec", 0, v, o)
        if "error" in c:
            assert out.startswith("rc -")
        else:
            assert out.startswith("rc 0") and [int(x) for x in np.frombuffer(o.read_bytes(), dtype="<u8")] == c["out"]
    # the K = 31 delta >= 2^60 case the reference dies on (tests/golden/g8_k31_delta_overflow.json)
    v.write_bytes(np.array([0, (1 << 62) - 1], dtype="<u8").tobytes())
    assert run(driver, "enc", 1, v, o).startswith("rc -")
    # empty inputs
    v.write_bytes(b"")
    assert run(driver, "enc", 1, v, o).startswith("rc 0") and o.read_bytes() == b""
    assert run(driver, "dec", 1, v, o).startswith("rc 0") and o.read_bytes() == b""


@pytest.mark.parametrize("chunk", [1, 7, 64, 1 << 20])
def test_parsers_under_sanitizers(driver, tmp_path, chunk):
    fq = G.load_json("g9_edge_fastq")["fastq"]
    fa = G.load_json("g9_edge_fasta")["fasta"]
    t, o = tmp_path / "t", tmp_path / "o"
    t.write_text(fq)
    assert run(driver, "fastq", chunk, t, o).startswith("rc 0 records 6")
    assert o.read_bytes() == "".join(x + "\n" for x in G.fastq_seqs(fq)).encode()
    t.write_text(fa)
    assert run(driver, "fasta", chunk, t, o).startswith("rc 0 records 3")
    assert o.read_bytes() == "".join(x + "\n" for x in G.fasta_seqs(fa)).encode()
    big = G.synth_fastq(G.load_json("g3_kmerize_genome"))
    if chunk >= 7:
        t.write_text(big)
        assert run(driver, "fastq", chunk, t, o).startswith("rc 0 records 1500")
        assert o.read_bytes() == "".join(x + "\n" for x in G.fastq_seqs(big)).encode()
    # a final record cut short, a last line without its newline, CRLF, text before the first FASTA header
    t.write_bytes(b"@r1\nACGT\n+\nIIII\n@r2\nAC")
    assert run(driver, "fastq", chunk, t, o).startswith("rc 0 records 1") and o.read_bytes() == b"ACGT\n"
    t.write_bytes(b"junk\r\n>r1\r\nACGT\r\nAC\r\n\r\n>r2\r\n>r3\r\nGG")
    assert run(driver, "fasta", chunk, t, o).startswith("rc 0 records 3") and o.read_bytes() == b"ACGTAC\n\nGG\n"


def test_fuzz_under_sanitizers(driver):
    for seed in (1, 2, 3):
        assert "fuzz ok" in run(driver, "fuzz", seed, 300)
