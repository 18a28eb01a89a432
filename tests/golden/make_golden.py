#!/usr/bin/env python3
"""
Capture golden vectors from the reference (drtconway/zotmer at /root/reference).

Runs ONLY in the development container (the reference does not travel); its outputs --
inputs and expected outputs as plain data -- are committed next to this script and are what
the tests read.  Two sources, both the reference's own code executing:

 (1) the unmodified arithmetic modules zotmer/library/{basics,bits,misc,codec64}.py, imported
     from /root/reference under Python 3 with `builtins.xrange = range` (they contain no other
     Python-2-only construct);
 (2) for the command drivers (kmerize, merge, dist, trim), which are Python 2 source (print
     statements, generator .next(), text-mode binary files), a throw-away copy under /tmp is
     passed through the stdlib's lib2to3 and nine one-line bytes/str fixes, docopt is stubbed
     with a dict, and the commands are driven in-process.  The copy never enters the repo.

Usage:  python3 tests/golden/make_golden.py        (rewrites tests/golden/*.json, *.npz)
"""
import builtins
import contextlib
import hashlib
import io
import json
import os
import random
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
WORK = "/tmp/zot3"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from zotmer_amd import synth  # noqa: E402  (the build's own input generator)


# ------------------------------------------------------------------------------------------
# (1) primitives from the unmodified modules
# ------------------------------------------------------------------------------------------

def primitives():
    builtins.xrange = range
    sys.path.insert(0, REF)
    from zotmer.library import basics, bits, misc, codec64
    sys.path.pop(0)

    rng = random.Random(20261004)
    g = {}

    seqs = ["ACGTA", "ACNGTAC", "acgu", "ACG", "", "GATTACA" * 5, "ACGTNNACGTACGTRYACGT-ACGTACGTAC",
            "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT", "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA",
            "".join(rng.choice("ACGT") for _ in range(200)),
            "".join(rng.choice("ACGTacgtUuNn*") for _ in range(300))]
    kl = []
    for s in seqs:
        for k in (1, 3, 4, 12, 25, 30, 31, 32):
            for both in (False, True):
                kl.append(dict(k=k, seq=s, both=both, out=[int(x) for x in basics.kmersList(k, s, both)]))
    g["kmersList"] = kl

    xs = [0, 1, 27, 0x23c48f123c48f, 0x37b0dec37b0d, (1 << 50) - 1, (1 << 62) - 1, (1 << 64) - 1]
    xs += [rng.getrandbits(50) for _ in range(40)] + [rng.getrandbits(64) for _ in range(20)]
    g["rev"] = [[x, int(bits.rev(x))] for x in xs]
    g["popcnt"] = [[x, int(bits.popcnt(x))] for x in xs]
    g["ffs"] = [[x, int(bits.ffs(x))] for x in xs]
    g["rc"] = [[k, x & ((1 << (2 * k)) - 1), int(basics.rc(k, x & ((1 << (2 * k)) - 1)))]
               for k in (1, 4, 12, 24, 25, 30, 31, 32) for x in xs]
    g["murmer"] = [[x, s, int(basics.murmer(x, s))] for x in xs for s in (0, 3, 17, (1 << 64) - 1)]
    g["fnv"] = [[x, s, int(basics.fnv(x, s))] for x in xs for s in (0, 17)]
    g["can"] = [[k, x & ((1 << (2 * k)) - 1), int(basics.can(k, x & ((1 << (2 * k)) - 1)))]
                for k in (12, 25, 30) for x in xs]
    g["sub"] = [[s, p, x, bool(basics.sub(s, p, x))] for x in xs for s in (0, 3)
                for p in (0.0, 0.05, 0.8, 1.0, 4.0, 7.99, 8.0, 9.0)]
    g["ham"] = [[a, b, int(basics.ham(a, b))] for a, b in zip(xs[:-1], xs[1:])]
    g["lcp"] = [[25, a & ((1 << 50) - 1), b & ((1 << 50) - 1),
                 int(basics.lcp(25, a & ((1 << 50) - 1), b & ((1 << 50) - 1)))] for a, b in zip(xs[:-1], xs[1:])]
    g["kmer"] = [[s, basics.kmer(s)] for s in ("ACGT", "acgu", "ACNT", "", "GATTACA" * 4)]
    g["render"] = [[k, x & ((1 << (2 * k)) - 1), basics.render(k, x & ((1 << (2 * k)) - 1))]
                   for k in (4, 25, 32) for x in xs[:12]]

    # radix_sort: > 16384 items so the MSD path runs (misc.py:404-406)
    v = [rng.getrandbits(50) for _ in range(40000)]
    w = list(v)
    misc.radix_sort(50, w)
    g["radix_sort"] = dict(bits=50, seed="random.Random(20261004) after the draws above", n=len(v),
                           sha256_in=hashlib.sha256(np.array(v, dtype="<u8").tobytes()).hexdigest(),
                           sha256_out=hashlib.sha256(np.array(w, dtype="<u8").tobytes()).hexdigest(),
                           is_sorted=(w == sorted(v)))
    np.savez_compressed(os.path.join(HERE, "radix_sort_in.npz"), xs=np.array(v, dtype=np.uint64))

    # codec64: every interesting width, runs that cross the 6-per-word cap, the 2**60 error
    cases = []
    widths = [0, 1, 10, 11, 12, 13, 15, 16, 20, 21, 30, 31, 59, 60]
    for wd in widths:
        top = (1 << wd) - 1 if wd else 0
        for n in (1, 2, 5, 6, 7, 13):
            cases.append([top] * n)
            cases.append([rng.getrandbits(wd) if wd else 0 for _ in range(n)])
    cases.append([1, 2, 3, 1000, 5, 6, 7, 8, 9, 10, 1 << 59, 3])
    cases.append([1] * 14)
    cases.append([])
    cases.append([rng.getrandbits(rng.choice(widths)) for _ in range(500)])
    cases.append([int(0.5 + 10 * rng.expovariate(0.5)) for _ in range(2000)])   # tests/test_files.py:11-12 shape
    enc = []
    for c in cases:
        ws = [int(w) for w in codec64.encode(c)]
        assert codec64.decodeList(ws) == c
        st = codec64.encoder()
        acc = []
        st.write = acc.append
        for x in c:
            st.append(x)
        st.end()
        assert [int(w) for w in acc] == ws          # streaming encoder == generator (codec64.py:42-120)
        enc.append(dict(values=c, words=ws))
    g["codec64"] = enc
    err = []
    for c in ([1 << 60], [5, 1 << 60, 7], [(1 << 64) - 1]):
        try:
            ws = [int(w) for w in codec64.encode(c)]
            # no exception from the generator itself: the word no longer fits 64 bits, and
            # files.writeWords' struct.pack('Q') (files.py:65-83) is what then fails
            err.append(dict(values=c, error=None, words=ws, fits_u64=all(w < (1 << 64) for w in ws)))
        except Exception as e:  # IndexError via _lookup[0]
            err.append(dict(values=c, error=type(e).__name__))
    g["codec64_errors"] = err
    # decoder on tags the encoder never emits
    dec = []
    for tag in range(16):
        wv = (rng.getrandbits(60) << 4) | tag
        try:
            dec.append(dict(word=wv, out=[int(x) for x in codec64.decodeList([wv])]))
        except Exception as e:
            dec.append(dict(word=wv, error=type(e).__name__))
    g["codec64_decode_tags"] = dec

    with open(os.path.join(HERE, "primitives.json"), "w") as f:
        json.dump(g, f)
    print("primitives.json:", {k: len(v) if hasattr(v, "__len__") else v for k, v in g.items()})


# ------------------------------------------------------------------------------------------
# (2) command drivers through the /tmp copy
# ------------------------------------------------------------------------------------------

def build_derived():
    shutil.rmtree(WORK, ignore_errors=True)
    os.makedirs(WORK + "/stubs")
    os.makedirs(WORK + "/work")
    shutil.copytree(REF + "/zotmer", WORK + "/zotmer")
    subprocess.check_call(["chmod", "-R", "u+w", WORK])
    files = [WORK + "/zotmer/library/%s.py" % m for m in
             ("basics", "bits", "misc", "codec64", "files", "file", "kmers", "reads", "dist", "exceptions", "timer", "stats")]
    files += [WORK + "/zotmer/library/container/%s.py" % m for m in ("__init__", "casket", "std", "vectors")]
    files += [WORK + "/zotmer/commands/%s.py" % m for m in ("kmerize", "merge", "dist", "trim", "jaccard", "project", "sample", "hist", "dump", "info")]
    subprocess.check_call([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n"] + files,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def patch(path, pairs):
        s = open(path).read()
        for a, b in pairs:
            assert a in s, (path, a)
            s = s.replace(a, b)
        open(path, "w").write(s)

    patch(WORK + "/zotmer/library/container/casket.py", [
        ("        if mode == 'r':\n            self.fo = open(fn, mode)", "        if mode == 'r':\n            self.fo = open(fn, 'rb')"),
        ("        elif mode == 'w':\n            self.fo = open(fn, mode)", "        elif mode == 'w':\n            self.fo = open(fn, 'wb')"),
        ("with open(fn, 'r') as f:", "with open(fn, 'rb') as f:"),
        ("            return ''", "            return b''"),
        ("        l = len(data)\n        self.fo.write(data)", "        data = data.encode() if isinstance(data, str) else data\n        l = len(data)\n        self.fo.write(data)"),
        ("        w = json.dumps(self.toc)", "        w = json.dumps(self.toc).encode()"),
    ])
    patch(WORK + "/zotmer/library/files.py", [("with open(t, 'w') as f:", "with open(t, 'wb') as f:")])
    patch(WORK + "/zotmer/commands/kmerize.py", [("open(t, 'w') as cf", "open(t, 'wb') as cf")])
    patch(WORK + "/zotmer/commands/merge.py", [("open(t, 'w') as cf", "open(t, 'wb') as cf")])
    # the reads iterator prints a warning through sys without importing it (reads.py:101); harmless
    with open(WORK + "/stubs/docopt.py", "w") as f:
        f.write("_next = {}\n\ndef docopt(doc, argv=None, **kw):\n    return dict(_next)\n")
    with open(WORK + "/stubs/tqdm.py", "w") as f:
        f.write("def tqdm(*a, **k):\n    raise RuntimeError('not used')\n")
    sys.path.insert(0, WORK + "/stubs")
    sys.path.insert(0, WORK)
    for m in [m for m in sys.modules if m == "zotmer" or m.startswith("zotmer.")]:
        del sys.modules[m]


def run(cmd, opts):
    import docopt
    import importlib
    docopt._next = opts
    mod = importlib.import_module("zotmer.commands." + cmd)
    out = io.StringIO()
    err = io.StringIO()
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        mod.main([cmd])
    return out.getvalue(), err.getvalue()


def kz(k, out, inputs, **kw):
    o = {"<k>": str(k), "<output>": out, "<input>": list(inputs), "-m": None, "-C": None, "-D": None, "-S": None, "-v": False}
    o.update(kw)
    return run("kmerize", o)


def load_set(path):
    """Read a set back with the derived copy's own reader."""
    from zotmer.library.kmers import kmers
    from zotmer.library.files import readKmersAndCounts
    from zotmer.library.container.casket import casket
    with kmers(path, "r") as z:
        meta = dict(z.meta)
        pairs = list(readKmersAndCounts(z))
    with casket(path, "r") as z:
        raw_k = z.open("kmers").read()
        raw_c = z.open("counts").read()
        toc = {k: [list(p) for p in v] for k, v in z.toc.items()}
    km = np.array([p[0] for p in pairs], dtype=np.uint64)
    ct = np.array([p[1] for p in pairs], dtype=np.uint64)
    return meta, km, ct, raw_k, raw_c, toc


def save_case(name, meta, km, ct, raw_k, raw_c, toc, extra=None):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), kmers=km, counts=ct,
                        raw_kmers=np.frombuffer(raw_k, dtype=np.uint8), raw_counts=np.frombuffer(raw_c, dtype=np.uint8))
    d = dict(meta=meta, toc=toc, n=int(len(km)), sum_counts=int(ct.sum()) if len(ct) else 0)
    if extra:
        d.update(extra)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
    print(name, "unique", len(km), "instances", d["sum_counts"], "meta keys", sorted(meta))


def commands():
    build_derived()
    W = WORK + "/work/"
    N_THR = synth.frac32(0.0005)
    S_THR = synth.frac32(0.005)

    # G2: 200 uniform reads with a high N rate, K=25
    g2 = dict(seed=synth.DEFAULT_SEED, first=0, count=200, L=150, genome=0, sub_thr=0, n_thr=synth.frac32(0.01))
    open(W + "g2.fastq", "w").write(synth.fastq_text(**g2))
    kz(25, W + "g2.k25", [W + "g2.fastq"])
    save_case("g2_kmerize_uniformN", *load_set(W + "g2.k25"), extra=dict(K=25, synth=g2, input="fastq_text(**synth)"))

    # G3: 1500 genome-sampled reads (20 kbp genome: counts well above 1), default -m and -m 1 must agree
    g3 = dict(seed=synth.DEFAULT_SEED + 1, first=0, count=1500, L=150, genome=20000, sub_thr=S_THR, n_thr=N_THR)
    open(W + "g3.fastq", "w").write(synth.fastq_text(**g3))
    kz(25, W + "g3.k25", [W + "g3.fastq"])
    a = load_set(W + "g3.k25")
    kz(25, W + "g3m1.k25", [W + "g3.fastq"], **{"-m": "1"})
    b = load_set(W + "g3m1.k25")
    same = bool(np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[0]["acgt"] == b[0]["acgt"]
                and a[0]["hist"] == b[0]["hist"] and a[3] == b[3] and a[4] == b[4])
    save_case("g3_kmerize_genome", *a, extra=dict(K=25, synth=g3, input="fastq_text(**synth)", m1_spill_path_identical=same))

    # G3b: same reads at K=24 (even K: palindromes exist) and K=12 (tiny key space)
    for k in (24, 12):
        kz(k, W + "g3.k%d" % k, [W + "g3.fastq"])
        save_case("g3_kmerize_genome_k%d" % k, *load_set(W + "g3.k%d" % k), extra=dict(K=k, synth=g3, input="fastq_text(**synth)"))

    # G8: K = 31 on the same reads (62-bit keys); and a K=31 input whose first delta needs > 60 bits
    kz(31, W + "g3.k31", [W + "g3.fastq"])
    save_case("g8_kmerize_k31", *load_set(W + "g3.k31"), extra=dict(K=31, synth=g3, input="fastq_text(**synth)"))
    open(W + "g8bad.fastq", "w").write("@r0\n%s\n+\n%s\n" % ("T" * 31, "I" * 31))
    try:
        kz(31, W + "g8bad.k31", [W + "g8bad.fastq"])
        bad = None
    except Exception as e:
        bad = type(e).__name__
    with open(os.path.join(HERE, "g8_k31_delta_overflow.json"), "w") as f:
        json.dump(dict(K=31, fastq="@r0\\n" + "T" * 31 + "\\n+\\n" + "I" * 31 + "\\n", error=bad), f, indent=1)
    print("K=31 poly-T error:", bad)

    # G9: input edge cases (lower case, U, IUPAC, short read, empty read, spaces, no final newline); FASTA multi-line; two files
    fq = ("@a\nacgtacgtacgtacgtacgtacgtacgtacgtacgtacgtacgtacgtac\n+\nx\n"
          "@b\nACGUACGUACGUACGUACGUACGUACGUACGUACGUACGUACGU\n+\nx\n"
          "@c\nACGTACGTACGTACGTACGTACGTACGTRACGTACGTACGTACGTACGTACGTACGT-ACGTACGTACGTACGTACGTACGTACGTA\n+\nx\n"
          "@d\nACGT\n+\nx\n"
          "@e\n\n+\n\n"
          "@f\n  ACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCA  \n+\nx")
    open(W + "g9.fastq", "w").write(fq)
    kz(25, W + "g9.k25", [W + "g9.fastq"])
    save_case("g9_edge_fastq", *load_set(W + "g9.k25"), extra=dict(K=25, fastq=fq))
    rng = random.Random(7)
    fa_seq = "".join(rng.choice("ACGT") for _ in range(350))
    fa = ">chr1 test\n" + "\n".join(fa_seq[i:i + 60] for i in range(0, 350, 60)) + "\n>chr2\nACGTNACGT\n" + \
         ">chr3\n" + "".join(rng.choice("ACGT") for _ in range(80)) + "\n"
    open(W + "g9.fa", "w").write(fa)
    kz(25, W + "g9fa.k25", [W + "g9.fa"])
    save_case("g9_edge_fasta", *load_set(W + "g9fa.k25"), extra=dict(K=25, fasta=fa))
    kz(25, W + "g9two.k25", [W + "g9.fastq", W + "g9.fa"])
    save_case("g9_two_files", *load_set(W + "g9two.k25"), extra=dict(K=25, inputs=["g9_edge_fastq.fastq", "g9_edge_fasta.fasta"]))

    # G10: -D 0.8 -S 3 (murmer subsample; effective rate 0.1) and -C capture
    kz(25, W + "g10.k25", [W + "g3.fastq"], **{"-D": "0.8", "-S": "3"})
    save_case("g10_kmerize_D0.8_S3", *load_set(W + "g10.k25"), extra=dict(K=25, synth=g3, D=0.8, S=3))
    bait = ">bait\n" + synth.read_strings(synth.DEFAULT_SEED + 1, 0, 1, 150, genome=20000)[0] + "\n"
    open(W + "bait.fa", "w").write(bait)
    kz(25, W + "g10c.k25", [W + "g3.fastq"], **{"-C": W + "bait.fa"})
    save_case("g10_kmerize_capture", *load_set(W + "g10c.k25"), extra=dict(K=25, synth=g3, bait_fasta=bait))

    # G4: merge of 2 and 3 (and 4, 5) sets; G6 trim; G5 dist
    parts = []
    for s in range(5):
        gs = dict(seed=synth.DEFAULT_SEED + 1, first=300 * s, count=400, L=150, genome=20000, sub_thr=S_THR, n_thr=N_THR)
        open(W + "p%d.fastq" % s, "w").write(synth.fastq_text(**gs))
        kz(25, W + "p%d.k25" % s, [W + "p%d.fastq" % s])
        save_case("g4_part%d" % s, *load_set(W + "p%d.k25" % s), extra=dict(K=25, synth=gs))
        parts.append(W + "p%d.k25" % s)
    for n in (2, 3, 4, 5):
        run("merge", {"<output>": W + "m%d.k25" % n, "<input>": parts[:n]})
        save_case("g4_merge%d" % n, *load_set(W + "m%d.k25" % n), extra=dict(inputs=["g4_part%d" % i for i in range(n)]))

    o, e = run("trim", {"-c": "3", "-C": "0", "<output>": W + "t3.k25", "<input>": W + "g3.k25"})
    save_case("g6_trim_c3", *load_set(W + "t3.k25"), extra=dict(input="g3_kmerize_genome", c=3, C=0))
    o, e = run("trim", {"-c": "2", "-C": "9", "<output>": W + "t29.k25", "<input>": W + "g3.k25"})
    save_case("g6_trim_c2_C9", *load_set(W + "t29.k25"), extra=dict(input="g3_kmerize_genome", c=2, C=9))

    dist = {}
    o, e = run("dist", {"-M": ["*.qual"], "<k>": "25", "<input>": parts[:3]})
    dist["qual_k25"] = dict(args=dict(M=["*.qual"], k=25, inputs=["g4_part0", "g4_part1", "g4_part2"]), stdout=o)
    o, e = run("dist", {"-M": ["jaccard.qual"], "<k>": "12", "<input>": parts[:2]})
    dist["jaccard_k12"] = dict(args=dict(M=["jaccard.qual"], k=12, inputs=["g4_part0", "g4_part1"]), stdout=o)
    o, e = run("dist", {"-M": ["jaccard.qual", "och*.qual"], "<k>": "20", "<input>": [parts[0], W + "m3.k25"]})
    dist["mixed_k20"] = dict(args=dict(M=["jaccard.qual", "och*.qual"], k=20, inputs=["g4_part0", "g4_merge3"]), stdout=o)
    o, e = run("dist", {"-M": ["list"], "<k>": "25", "<input>": []})
    dist["list"] = dict(stdout=o)
    # raw (a, b, c) straight from the reference's split on the projected sets
    import zotmer.library.dist as rdist
    import importlib
    cd = importlib.import_module("zotmer.commands.dist")
    trip = {}
    for k in (25, 20, 12):
        lhs = cd.measures["jaccard.qual"].prep(k, parts[0])
        rhs = cd.measures["jaccard.qual"].prep(k, parts[1])
        trip[str(k)] = dict(abc=[int(v) for v in rdist.split(lhs, rhs)], nx=len(lhs), ny=len(rhs))
    dist["split_part0_part1"] = trip
    with open(os.path.join(HERE, "g5_dist.json"), "w") as f:
        json.dump(dist, f, indent=1, sort_keys=True)
    print("dist:", {k: (v.get("stdout", "") or "")[:60] for k, v in dist.items() if isinstance(v, dict)})

    # f3: jaccard / project / sample on the same sets
    f3 = {}
    o, e = run("jaccard", {"-a": False, "-b": False, "-p": None, "<input>": parts[:3]})
    f3["jaccard_default"] = dict(inputs=["g4_part0", "g4_part1", "g4_part2"], stdout=o)
    o, e = run("jaccard", {"-a": True, "-b": False, "-p": None, "<input>": parts[:3]})
    f3["jaccard_all"] = dict(inputs=["g4_part0", "g4_part1", "g4_part2"], stdout=o)
    o, e = run("jaccard", {"-a": False, "-b": False, "-p": "0.5", "<input>": parts[:2]})
    f3["jaccard_p0.5"] = dict(inputs=["g4_part0", "g4_part1"], stdout=o)
    jfa = ">s1 first\n" + synth.read_strings(synth.DEFAULT_SEED + 1, 0, 1, 150, genome=20000)[0] + "\n" + \
          ">s2\n" + synth.read_strings(synth.DEFAULT_SEED + 1, 0, 1, 150, genome=20000)[0][20:] + "ACGTTGCA\n" + \
          ">s3 x\n" + synth.read_strings(synth.DEFAULT_SEED + 1, 7, 1, 150, genome=20000)[0] + "\n"
    open(W + "j.fa", "w").write(jfa)
    o, e = run("jaccard", {"-a": True, "-b": False, "-p": None, "<input>": [W + "j.fa"]})
    f3["jaccard_fasta_all"] = dict(fasta=jfa, stdout=o)
    with open(os.path.join(HERE, "f3_jaccard.json"), "w") as f:
        json.dump(f3, f, indent=1, sort_keys=True)
    run("project", {"<ref>": parts[1], "<output>": W + "prj.k25", "<input>": parts[0]})
    save_case("f3_project_part0_on_part1", *load_set(W + "prj.k25"), extra=dict(ref="g4_part1", input="g4_part0"))
    run("sample", {"-D": True, "-S": "5", "-P": "0.3", "<output>": W + "smp.k25", "<input>": parts[0]})
    save_case("f3_sample_D_S5_P0.3", *load_set(W + "smp.k25"), extra=dict(input="g4_part0", S=5, P=0.3))
    run("sample", {"-D": False, "-S": None, "-P": None, "<output>": W + "smp2.k25", "<input>": parts[0]})
    save_case("f3_sample_defaults", *load_set(W + "smp2.k25"), extra=dict(input="g4_part0", S=0, P=0.01,
              note="docopt gives False (not None) for an absent -D, so commands/sample.py:51 always takes the deterministic branch"))
    print("f3:", {k: v["stdout"][:70] for k, v in f3.items()})

    # f4: the inspection commands' stdout (hist / dump / info) on two of the sets above.  The reference prints the path it
    # was given; it is replaced by the case name.  NB `zot info` prints Python reprs of the JSON-loaded metadata: under the
    # reference's Python 2 strings carry a u'' prefix and dict order is arbitrary; this is the Python 3 rendering of the
    # same code (sorted by key at the top level, insertion = file order inside `hist`).
    f4 = {}
    for case, path in (("g9_edge_fastq", W + "g9.k25"), ("g4_merge3", W + "m3.k25")):
        for cmd in ("hist", "dump", "info"):
            o, e = run(cmd, {"<input>": path if cmd == "dump" else [path]})
            o = o.replace(path, case)
            if len(o) > (1 << 16):      # a long dump is kept as its digest, line count and first lines
                f4["%s_%s" % (cmd, case)] = dict(input=case, sha256=hashlib.sha256(o.encode()).hexdigest(), lines=o.count("\n"),
                                                 head="".join(o.splitlines(True)[:20]), stderr=e)
            else:
                f4["%s_%s" % (cmd, case)] = dict(input=case, stdout=o, stderr=e)
    with open(os.path.join(HERE, "f4_inspect.json"), "w") as f:
        json.dump(f4, f, indent=1, sort_keys=True)
    print("f4:", {k: len(v.get("stdout", v.get("head"))) for k, v in f4.items()})

    # -C together with -D: the reference's `if d is not None: ... elif B is not None:` (kmerize.py:494-520) ignores the baits
    kz(25, W + "g10cd.k25", [W + "g3.fastq"], **{"-C": W + "bait.fa", "-D": "0.8", "-S": "3"})
    save_case("g10_kmerize_capture_and_D", *load_set(W + "g10cd.k25"), extra=dict(K=25, synth=g3, bait_fasta=bait, D=0.8, S=3))

    # config 1 at full size (10 000 x 150 bp genome-sampled): digests only
    c = synth.CONFIGS["config1"]
    g1 = dict(seed=synth.DEFAULT_SEED, first=0, count=c["reads"], L=c["L"], genome=c["genome"],
              sub_thr=synth.frac32(c["sub"]), n_thr=synth.frac32(c["n"]))
    open(W + "c1.fastq", "w").write(synth.fastq_text(**g1))
    kz(25, W + "c1.k25", [W + "c1.fastq"])
    meta, km, ct, rk, rc_, toc = load_set(W + "c1.k25")
    with open(os.path.join(HERE, "config1_digest.json"), "w") as f:
        json.dump(dict(K=25, synth=g1, meta=meta, n=int(len(km)), sum_counts=int(ct.sum()),
                       sha256_kmers=hashlib.sha256(km.astype("<u8").tobytes()).hexdigest(),
                       sha256_counts=hashlib.sha256(ct.astype("<u8").tobytes()).hexdigest(),
                       sha256_raw_kmers=hashlib.sha256(rk).hexdigest(), sha256_raw_counts=hashlib.sha256(rc_).hexdigest(),
                       len_raw_kmers=len(rk), len_raw_counts=len(rc_)), f, indent=1, sort_keys=True)
    print("config1: unique", len(km), "instances", int(ct.sum()))


if __name__ == "__main__":
    primitives()
    if "--primitives-only" not in sys.argv:
        commands()
    shutil.rmtree(WORK, ignore_errors=True)
