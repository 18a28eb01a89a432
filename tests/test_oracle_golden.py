"""
The CPU oracle (oracle/zk_oracle.c) against the vectors captured from the reference.
This is what pins the oracle; the GPU parity tests then compare the HIP path with it.
"""
import fnmatch
import hashlib

import numpy as np
import pytest

from oracle import zkoracle as zo
from tests import _golden as G

P = G.load_json("primitives")


def test_kmers_list():
    for c in P["kmersList"]:
        got = zo.kmers_list(c["k"], c["seq"], c["both"])
        assert [int(x) for x in got] == c["out"], (c["k"], c["seq"], c["both"])


def test_bit_primitives():
    for x, r in P["rev"]:
        assert zo.rev(x) == r
    for x, r in P["popcnt"]:
        assert zo.popcnt(x) == r
    for x, r in P["ffs"]:
        assert zo.ffs(x) == r
    for k, x, r in P["rc"]:
        assert zo.rc(k, x) == r
    for a, b, r in P["ham"]:
        assert zo.ham(a, b) == r
    for k, a, b, r in P["lcp"]:
        assert zo.lcp(k, a, b) == r


def test_hashes():
    for x, s, r in P["murmer"]:
        assert zo.murmer(x, s) == r
    for x, s, r in P["fnv"]:
        assert zo.fnv(x, s) == r
    for k, x, r in P["can"]:
        assert zo.can(k, x) == r
    for s, p, x, r in P["sub"]:
        assert zo.sub(s, p, x) == r, (s, p, x)


def test_kmer_render():
    for s, r in P["kmer"]:
        assert zo.kmer(s) == r
    for k, x, r in P["render"]:
        assert zo.render(k, x) == r


def test_radix_sort():
    g = P["radix_sort"]
    xs = np.load(G.GOLD + "/radix_sort_in.npz")["xs"]
    assert hashlib.sha256(xs.astype("<u8").tobytes()).hexdigest() == g["sha256_in"]
    out = zo.radix_sort(g["bits"], xs)
    assert hashlib.sha256(out.astype("<u8").tobytes()).hexdigest() == g["sha256_out"]


def test_codec64_words():
    for c in P["codec64"]:
        w = zo.codec64_encode(c["values"])
        assert [int(x) for x in w] == c["words"]
        assert [int(x) for x in zo.codec64_decode(c["words"])] == c["values"]
    for c in P["codec64_errors"]:
        # the reference either raises in the encoder or emits a word wider than 64 bits that
        # its struct.pack('Q') writer then rejects; either way the value has no on-disk form
        assert c["error"] is not None or not c["fits_u64"]
        with pytest.raises(IndexError):
            zo.codec64_encode(c["values"])
    for c in P["codec64_decode_tags"]:
        if "error" in c:
            with pytest.raises(KeyError):
                zo.codec64_decode([c["word"]])
        else:
            assert [int(x) for x in zo.codec64_decode([c["word"]])] == c["out"]


def _check_set(info, km, ct, r, with_hist=True):
    assert np.array_equal(r["kmers"], km)
    assert np.array_equal(r["counts"].astype(np.uint64), ct)
    meta = info["meta"]
    assert r["reads"] == meta["reads"]
    tot = float(sum(r["acgt"]))
    assert [c / tot for c in r["acgt"]] == meta["acgt"]          # kmerize.py:554-555
    if with_hist:
        assert G.hist_dict(r["counts"]) == meta["hist"]


@pytest.mark.parametrize("name", G.KMERIZE_SYNTH_CASES)
def test_kmerize_synth(name):
    info, km, ct, raw_k, raw_c = G.load_case(name)
    reads = G.synth_reads(info)
    r = zo.kmerize(info["K"], reads)
    _check_set(info, km, ct, r)
    # flush boundaries do not change the result (the reference's -m 1 run was byte-identical)
    r2 = zo.kmerize(info["K"], reads, flush_at=5000)
    assert np.array_equal(r2["kmers"], km) and np.array_equal(r2["counts"].astype(np.uint64), ct)
    # the two on-disk streams: delta + codec64 of the k-mers, codec64 of the counts
    assert zo.codec64_encode(zo.delta(km)).astype("<u8").tobytes() == raw_k
    assert zo.codec64_encode(ct).astype("<u8").tobytes() == raw_c
    assert np.array_equal(zo.undelta(zo.codec64_decode(np.frombuffer(raw_k, dtype="<u8"))), km)


def test_kmerize_edge_inputs():
    info, km, ct, _, _ = G.load_case("g9_edge_fastq")
    fq = G.fastq_seqs(info["fastq"])
    assert len(fq) == 6
    _check_set(info, km, ct, zo.kmerize(25, fq))
    info2, km2, ct2, _, _ = G.load_case("g9_edge_fasta")
    fa = G.fasta_seqs(info2["fasta"])
    assert len(fa) == 3 and len(fa[0]) == 350
    _check_set(info2, km2, ct2, zo.kmerize(25, fa))
    info3, km3, ct3, _, _ = G.load_case("g9_two_files")
    _check_set(info3, km3, ct3, zo.kmerize(25, fq + fa))


def test_kmerize_subsample_and_capture():
    info, km, ct, _, _ = G.load_case("g10_kmerize_D0.8_S3")
    reads = G.synth_reads(info)
    r = zo.kmerize(25, reads, mode=1, p=info["D"], seed=info["S"])
    _check_set(info, km, ct, r)
    info, km, ct, _, _ = G.load_case("g10_kmerize_capture")
    bait = G.fasta_seqs(info["bait_fasta"])
    b = np.unique(np.concatenate([zo.kmers_list(25, s, True) for s in bait]))
    r = zo.kmerize(25, reads, mode=2, baits=b)
    _check_set(info, km, ct, r)
    # -C with -D: the reference takes the -D branch and never looks at the baits (kmerize.py:494-520)
    info, km, ct, _, _ = G.load_case("g10_kmerize_capture_and_D")
    _check_set(info, km, ct, zo.kmerize(25, reads, mode=1, p=info["D"], seed=info["S"]))


def test_k31_delta_overflow():
    g = G.load_json("g8_k31_delta_overflow")
    assert g["error"] is not None
    r = zo.kmerize(31, ["T" * 31])
    with pytest.raises(IndexError):
        zo.codec64_encode(zo.delta(r["kmers"]))


def test_merge():
    parts = [G.load_case("g4_part%d" % i) for i in range(5)]
    for n in (2, 3, 4, 5):
        info, km, ct, raw_k, raw_c = G.load_case("g4_merge%d" % n)
        zs, zc, acgt = zo.merge_n(25, [(p[1], p[2]) for p in parts[:n]])
        assert np.array_equal(zs, km) and np.array_equal(zc, ct)
        assert G.hist_dict(zc) == info["meta"]["hist"]
        if n > 2:   # mergeNinto path: count-weighted acgt (merge.py:159)
            tot = float(sum(acgt))
            assert [c / tot for c in acgt] == info["meta"]["acgt"]
        else:       # pairwise path: hist() counts distinct k-mers (merge.py:88-92)
            a = [int(np.sum((zs & np.uint64(3)) == np.uint64(b))) for b in range(4)]
            assert [c / float(sum(a)) for c in a] == info["meta"]["acgt"]
        assert zo.codec64_encode(zo.delta(zs)).astype("<u8").tobytes() == raw_k
        assert zo.codec64_encode(zc).astype("<u8").tobytes() == raw_c
    # 2-way streaming merge equals the k-way one
    zs, zc = zo.union_sum(parts[0][1], parts[0][2], parts[1][1], parts[1][2])
    _, km, ct, _, _ = G.load_case("g4_merge2")
    assert np.array_equal(zs, km) and np.array_equal(zc, ct)


def test_trim():
    _, km, ct, _, _ = G.load_case("g3_kmerize_genome")
    for name in ("g6_trim_c3", "g6_trim_c2_C9"):
        info, tk, tc, _, _ = G.load_case(name)
        ox, oc = zo.trim(km, ct, info["c"], info["C"])
        assert np.array_equal(ox, tk) and np.array_equal(oc, tc)


def _dist_lines(measures, pairs_abc, names):
    out = ["\t".join(["lhs.name", "rhs.name"] + measures)]
    for (i, j), abc in pairs_abc:
        out.append("\t".join([names[i], names[j]] + ["%g" % zo.QUAL_MEASURES[m](*abc) for m in measures]))
    return "\n".join(out) + "\n"


def test_dist():
    g = G.load_json("g5_dist")
    sets = {n: G.load_case(n)[1] for n in ("g4_part0", "g4_part1", "g4_part2", "g4_merge3")}
    for k, t in g["split_part0_part1"].items():
        sh = 2 * (25 - int(k))
        x = zo.project_dedupe(sets["g4_part0"], sh)
        y = zo.project_dedupe(sets["g4_part1"], sh)
        assert (len(x), len(y)) == (t["nx"], t["ny"])
        assert list(zo.split(x, y)) == t["abc"]
    for key in ("qual_k25", "jaccard_k12", "mixed_k20"):
        a = g[key]["args"]
        ms = sorted(m for m in zo.QUAL_MEASURES if any(fnmatch.fnmatch(m, pat) for pat in a["M"]))
        sh = 2 * (25 - a["k"])
        prj = [zo.project_dedupe(sets[n], sh) for n in a["inputs"]]
        pairs = [((i, j), zo.split(prj[i], prj[j])) for i in range(len(prj)) for j in range(i + 1, len(prj))]
        ref = g[key]["stdout"]
        names = []     # the path names the reference printed, in input order
        for row in (l.split("\t") for l in ref.strip().split("\n")[1:]):
            for nm in row[:2]:
                if nm not in names:
                    names.append(nm)
        assert _dist_lines(ms, pairs, names) == ref


def test_config1_digest():
    g = G.load_json("config1_digest")
    reads = G.synth_reads(g)
    r = zo.kmerize(g["K"], reads)
    assert len(r["kmers"]) == g["n"] and int(r["counts"].sum()) == g["sum_counts"]
    assert hashlib.sha256(r["kmers"].astype("<u8").tobytes()).hexdigest() == g["sha256_kmers"]
    assert hashlib.sha256(r["counts"].astype("<u8").tobytes()).hexdigest() == g["sha256_counts"]
    rk = zo.codec64_encode(zo.delta(r["kmers"])).astype("<u8").tobytes()
    rc = zo.codec64_encode(r["counts"]).astype("<u8").tobytes()
    assert (len(rk), len(rc)) == (g["len_raw_kmers"], g["len_raw_counts"])
    assert hashlib.sha256(rk).hexdigest() == g["sha256_raw_kmers"]
    assert hashlib.sha256(rc).hexdigest() == g["sha256_raw_counts"]
    assert G.hist_dict(r["counts"]) == g["meta"]["hist"]
    tot = float(sum(r["acgt"]))
    assert [c / tot for c in r["acgt"]] == g["meta"]["acgt"]


def test_python_restatement_config1_digest():
    """oracle/py_restatement.py (the pure-Python 'reference CPU path' bench.py times) on config 1
    against what the reference itself produced: both encoded streams byte for byte (sha256), hist, acgt, reads."""
    import struct
    from oracle import py_restatement as pr
    g = G.load_json("config1_digest")
    r = pr.kmerize(g["K"], G.synth_reads(g))
    assert len(r["kmers"]) == g["n"] and sum(r["counts"]) == g["sum_counts"] and r["instances"] == g["sum_counts"]
    assert hashlib.sha256(struct.pack("<%dQ" % len(r["kmers"]), *r["kmers"])).hexdigest() == g["sha256_kmers"]
    assert hashlib.sha256(r["kmers_bytes"]).hexdigest() == g["sha256_raw_kmers"]
    assert hashlib.sha256(r["counts_bytes"]).hexdigest() == g["sha256_raw_counts"]
    assert (len(r["kmers_bytes"]), len(r["counts_bytes"])) == (g["len_raw_kmers"], g["len_raw_counts"])
    assert {str(k): v for k, v in r["hist"].items()} == g["meta"]["hist"]
    assert r["acgt"] == g["meta"]["acgt"] and r["reads"] == g["meta"]["reads"]
    # the flush path (KmerAccumulator2 merges a new sorted buffer into the table) gives the same arrays
    r2 = pr.kmerize(g["K"], G.synth_reads(g)[:3000], flush_at=100_000)
    r1 = pr.kmerize(g["K"], G.synth_reads(g)[:3000])
    assert r1["kmers"] == r2["kmers"] and r1["counts"] == r2["counts"]


# ---- next-row commands (f3): jaccard / project / sample ------------------------------------------------

def _rename(ref, names_new):
    """The reference printed its temporary paths; map them to given names in order of appearance."""
    names = []
    for row in (l.split("\t") for l in ref.strip().split("\n")):
        for nm in row[:2]:
            if nm not in names and "\t" not in nm and "/" in nm:
                names.append(nm)
    for old, new in zip(names, names_new):
        ref = ref.replace(old, new)
    return ref


def test_jaccard_lines():
    g = G.load_json("f3_jaccard")
    sets = {n: G.load_case(n)[1] for n in ("g4_part0", "g4_part1", "g4_part2")}
    for key, pairs, p in (("jaccard_default", [(0, 1), (0, 2)], None), ("jaccard_all", [(0, 1), (0, 2), (1, 2)], None),
                          ("jaccard_p0.5", [(0, 1)], 0.5)):
        names = g[key]["inputs"]
        want = _rename(g[key]["stdout"], names)
        got = ""
        for i, j in pairs:
            a, b, c = zo.split(sets[names[i]], sets[names[j]])
            got += zo.jaccard_line(names[i], names[j], a + b, a + c, a, p) + "\n"
        assert got == want
    # FASTA mode: K = 25, both-strand k-mer set of every record (jaccard.py:100-125)
    fa = g["jaccard_fasta_all"]
    recs = [(l[1:].split()[0]) for l in fa["fasta"].split("\n") if l.startswith(">")]
    seqs = G.fasta_seqs(fa["fasta"])
    ks = [np.unique(zo.kmers_list(25, s, True)) for s in seqs]
    got = "%d\n" % len(seqs)
    for i in range(len(seqs)):
        for j in range(i + 1, len(seqs)):
            a, b, c = zo.split(ks[i], ks[j])
            got += zo.jaccard_line(recs[i], recs[j], len(ks[i]), len(ks[j]), a) + "\n"
    assert got == fa["stdout"]


def test_project_and_sample():
    _, k0, c0, _, _ = G.load_case("g4_part0")
    _, k1, _, _, _ = G.load_case("g4_part1")
    info, pk, pc, _, _ = G.load_case("f3_project_part0_on_part1")
    ok, oc = zo.project(k1, k0, c0)
    assert np.array_equal(ok, pk) and np.array_equal(oc, pc)
    for name in ("f3_sample_D_S5_P0.3", "f3_sample_defaults"):
        info, sk, sc, _, _ = G.load_case(name)
        ok, oc = zo.sample_d(info["P"], info["S"], k0, c0)
        assert np.array_equal(ok, sk) and np.array_equal(oc, sc)
        assert G.hist_dict(oc) == info["meta"]["hist"]
