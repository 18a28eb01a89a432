"""
End-to-end `zot <command>` on the GPU against the files the reference wrote for the same inputs
(tests/golden): the two codec64 payload streams must match byte for byte, the metadata value for
value, and `zot dist` stdout character for character.
"""
import gzip
import io
import json
import os
import struct
from contextlib import redirect_stdout

import numpy as np
import pytest

from tests import _golden as G
from zotmer_amd import cli
from zotmer_amd.library import vectors
from zotmer_amd.library.container import Container, KmerSet

pytestmark = pytest.mark.gpu


def zot(*args):
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = cli.main_inner([str(a) for a in args])
    return buf.getvalue(), rc


def members(path):
    with Container(str(path), "r") as z:
        return {nm: z.read(nm) for nm, _ in z.names()}


def check_file(path, name, meta_keys=("K", "hist", "acgt", "reads")):
    info, km, ct, raw_k, raw_c = G.load_case(name)
    m = members(path)
    assert m["kmers"] == raw_k, "kmers stream differs from the reference's bytes"
    assert m["counts"] == raw_c, "counts stream differs from the reference's bytes"
    meta = json.loads(m["__meta__"].decode())
    for k in meta_keys:
        if k in info["meta"]:
            assert meta[k] == info["meta"][k], k
    assert meta["kmers"] == "kmers" and meta["counts"] == "counts"
    return meta


def write_case_fastq(tmp_path, name, fname="in.fastq"):
    info = G.load_json(name)
    p = tmp_path / fname
    p.write_text(G.synth_fastq(info))
    return info, p


@pytest.mark.parametrize("name", G.KMERIZE_SYNTH_CASES)
def test_kmerize_files(tmp_path, name):
    info, fq = write_case_fastq(tmp_path, name)
    out = tmp_path / "out.k"
    zot("kmerize", info["K"], out, fq)
    check_file(out, name)
    # BASELINE.json's spelling, small device batches (several union-sum rounds), gz input
    gz = tmp_path / "in2.fastq.gz"
    with gzip.open(str(gz), "wb") as f:
        f.write(fq.read_bytes())
    out2 = tmp_path / "out2.k"
    zot("kmerize", "-k", info["K"], "-m", "1", out2, gz)
    assert members(out2)["kmers"] == members(out)["kmers"] and members(out2)["counts"] == members(out)["counts"]
    check_file(out2, name)


def test_kmerize_edge_inputs(tmp_path):
    fq = tmp_path / "e.fastq"
    fa = tmp_path / "e.fa"
    fq.write_text(G.load_json("g9_edge_fastq")["fastq"])
    fa.write_text(G.load_json("g9_edge_fasta")["fasta"])
    zot("kmerize", 25, tmp_path / "a.k25", fq)
    check_file(tmp_path / "a.k25", "g9_edge_fastq")
    zot("kmerize", 25, tmp_path / "b.k25", fa)
    check_file(tmp_path / "b.k25", "g9_edge_fasta")
    zot("kmerize", 25, tmp_path / "c.k25", fq, fa)
    check_file(tmp_path / "c.k25", "g9_two_files")


def test_kmerize_subsample_and_capture(tmp_path):
    info, fq = write_case_fastq(tmp_path, "g10_kmerize_D0.8_S3")
    zot("kmerize", "-D", "0.8", "-S", "3", 25, tmp_path / "d.k25", fq)
    check_file(tmp_path / "d.k25", "g10_kmerize_D0.8_S3")
    cinfo = G.load_json("g10_kmerize_capture")
    bait = tmp_path / "bait.fa"
    bait.write_text(cinfo["bait_fasta"])
    zot("kmerize", "-C", bait, 25, tmp_path / "c.k25", fq)
    check_file(tmp_path / "c.k25", "g10_kmerize_capture")
    zot("kmerize", "-C", bait, "-m", "1", 25, tmp_path / "c2.k25", fq)
    check_file(tmp_path / "c2.k25", "g10_kmerize_capture")
    # -C together with -D: the reference's `if d is not None ... elif B is not None` lets -D win and ignores the baits
    zot("kmerize", "-C", bait, "-D", "0.8", "-S", "3", 25, tmp_path / "cd.k25", fq)
    check_file(tmp_path / "cd.k25", "g10_kmerize_capture_and_D")
    gcd = G.load_case("g10_kmerize_capture_and_D")
    gd = G.load_case("g10_kmerize_D0.8_S3")
    assert np.array_equal(gcd[1], gd[1]) and np.array_equal(gcd[2], gd[2])      # (the two reference outputs are the same set)


def test_kmerize_k31_overflow_raises(tmp_path):
    fq = tmp_path / "t.fastq"
    fq.write_text("@r0\n%s\n+\n%s\n" % ("T" * 31, "I" * 31))
    with pytest.raises(vectors.CodecError):          # the reference dies here too (golden g8_k31_delta_overflow)
        zot("kmerize", 31, tmp_path / "t.k31", fq)


def make_set(tmp_path, name):
    """Materialise a golden set as a container file."""
    info, km, ct, _, _ = G.load_case(name)
    p = tmp_path / (name + ".k25")
    with KmerSet(str(p), "w") as z:
        vectors.write_kmers_and_counts(z, km, ct)
        z.meta.update(info["meta"])
        z.meta.setdefault("K", 25)
    return p


def test_merge_files(tmp_path):
    parts = [make_set(tmp_path, "g4_part%d" % i) for i in range(5)]
    for n in (2, 3, 4, 5):
        out = tmp_path / ("m%d.k25" % n)
        zot("merge", out, *parts[:n])
        meta = check_file(out, "g4_merge%d" % n, meta_keys=("hist",) if n <= 2 else ("K", "hist", "acgt"))
        assert meta["K"] == 25 and "reads" not in meta
    # mismatched K: message + exit status 1 (merge.py:186-190)
    other = tmp_path / "k24.k24"
    info, km, ct, _, _ = G.load_case("g3_kmerize_genome_k24")
    with KmerSet(str(other), "w") as z:
        vectors.write_kmers_and_counts(z, km, ct)
        z.meta.update(info["meta"])
    with pytest.raises(SystemExit) as e:
        zot("merge", tmp_path / "bad.k", parts[0], other)
    assert e.value.code == 1


def test_dist_stdout(tmp_path):
    g = G.load_json("g5_dist")
    files = {n: make_set(tmp_path, n) for n in ("g4_part0", "g4_part1", "g4_part2", "g4_merge3")}
    for key in ("qual_k25", "jaccard_k12", "mixed_k20"):
        a = g[key]["args"]
        argv = ["dist"]
        for m in a["M"]:
            argv += ["-M", m]
        argv += [a["k"]] + [files[n] for n in a["inputs"]]
        out, _ = zot(*argv)
        ref = g[key]["stdout"]
        # the reference printed its own temporary paths: map them to ours, in order of appearance
        names = []
        for row in (l.split("\t") for l in ref.strip().split("\n")[1:]):
            for nm in row[:2]:
                if nm not in names:
                    names.append(nm)
        for old, new in zip(names, [str(files[n]) for n in a["inputs"]]):
            ref = ref.replace(old, new)
        assert out == ref
    out, _ = zot("dist", "-M", "list", 25)
    assert out == g["list"]["stdout"]
    out, _ = zot("dist", 25, files["g4_part0"], files["g4_part1"])       # no -M: prints nothing (dist.py:120-124)
    assert out == ""


def test_trim_files(tmp_path):
    src = make_set(tmp_path, "g3_kmerize_genome")
    zot("trim", "-c", 3, tmp_path / "t3.k25", src)
    meta = check_file(tmp_path / "t3.k25", "g6_trim_c3")
    assert meta["hist"] == G.load_json("g3_kmerize_genome")["meta"]["hist"]          # copied, not recomputed (trim.py:95)
    zot("trim", "-c", 2, "-C", 9, tmp_path / "t29.k25", src)
    check_file(tmp_path / "t29.k25", "g6_trim_c2_C9")


def test_info_hist_dump(tmp_path):
    """stdout of the inspection commands against what the reference's own hist.py / dump.py / info.py printed
    (tests/golden/f4_inspect.json, captured by make_golden.py; the path the reference printed is the case name there)."""
    import ast
    import hashlib
    g = G.load_json("f4_inspect")
    for case in ("g9_edge_fastq", "g4_merge3"):
        src = make_set(tmp_path, case)
        out, _ = zot("hist", src)
        assert out.replace(str(src), case) == g["hist_" + case]["stdout"]
        out, _ = zot("dump", src)
        want = g["dump_" + case]
        if "stdout" in want:
            assert out == want["stdout"]
        else:
            assert hashlib.sha256(out.encode()).hexdigest() == want["sha256"] and out.count("\n") == want["lines"]
            assert out.startswith(want["head"])
        # info prints Python reprs of the metadata; the order INSIDE the hist dict is the file's JSON order, which not even the
        # reference reproduces from run to run, so that one value is compared as a dict
        out, _ = zot("info", src)
        got = dict(l.split(" ", 1) for l in out.strip().split("\n"))
        ref = dict(l.split(" ", 1) for l in g["info_" + case]["stdout"].strip().split("\n"))
        assert sorted(got) == sorted(ref) and list(got) == sorted(got)
        for k in ref:
            if k == "hist":
                assert ast.literal_eval(got[k]) == ast.literal_eval(ref[k])
            else:
                assert got[k] == ref[k], k
    _, rc = zot("nosuchcommand")
    assert rc == 1


def _retarget(ref, paths):
    """the reference printed its own temporary paths: map them to ours, in order of appearance"""
    names = []
    for row in (l.split("\t") for l in ref.strip().split("\n")):
        for nm in row[:2]:
            if nm not in names and "/" in nm:
                names.append(nm)
    for old, new in zip(names, paths):
        ref = ref.replace(old, new)
    return ref


def test_jaccard_stdout(tmp_path):
    g = G.load_json("f3_jaccard")
    files = {n: make_set(tmp_path, n) for n in ("g4_part0", "g4_part1", "g4_part2")}
    for key, flags in (("jaccard_default", []), ("jaccard_all", ["-a"]), ("jaccard_p0.5", ["-p", "0.5"])):
        paths = [str(files[n]) for n in g[key]["inputs"]]
        out, _ = zot("jaccard", *(flags + paths))
        assert out == _retarget(g[key]["stdout"], paths), key
    fa = tmp_path / "two.fa"
    fa.write_text(g["jaccard_fasta_all"]["fasta"])
    out, _ = zot("jaccard", "-a", fa)
    assert out == g["jaccard_fasta_all"]["stdout"]
    # mismatched K: message + exit status 1 (jaccard.py:151-153)
    other = tmp_path / "k24.k24"
    info, km, ct, _, _ = G.load_case("g3_kmerize_genome_k24")
    with KmerSet(str(other), "w") as z:
        vectors.write_kmers_and_counts(z, km, ct)
        z.meta.update(info["meta"])
    with pytest.raises(SystemExit) as e:
        zot("jaccard", files["g4_part0"], other)
    assert e.value.code == 1


def test_project_and_sample_files(tmp_path):
    p0, p1 = make_set(tmp_path, "g4_part0"), make_set(tmp_path, "g4_part1")
    out = tmp_path / "proj.k25"
    zot("project", p1, out, p0)
    meta = check_file(out, "f3_project_part0_on_part1", meta_keys=("K", "hist"))
    assert "acgt" not in meta and "reads" not in meta                  # only K and hist travel (project.py:55-66)
    out = tmp_path / "s1.k25"
    zot("sample", "-D", "-S", 5, "-P", 0.3, out, p0)
    check_file(out, "f3_sample_D_S5_P0.3")
    out = tmp_path / "s2.k25"
    zot("sample", out, p0)                                             # defaults: P 0.01, seed 0, still hash-based
    check_file(out, "f3_sample_defaults")
    # an input without counts projects to an output without counts
    info, km, ct, _, _ = G.load_case("g4_part0")
    bare = tmp_path / "bare.k25"
    with KmerSet(str(bare), "w") as z:
        z.add("kmers", vectors.encode_kmers(km))
        z.meta.update({"K": 25, "kmers": "kmers", "hist": {}})
    out = tmp_path / "proj2.k25"
    zot("project", p1, out, bare)
    m = members(out)
    assert m["kmers"] == G.load_case("f3_project_part0_on_part1")[3] and "counts" not in m
