#!/usr/bin/env python3
"""End-to-end `zot kmerize` / `merge` / `dist` / `trim` on files (run on the GPU box): where does the wall
time go once the kernels are fast?  Writes a synthetic FASTQ of N reads to /tmp first."""
import io, json, os, sys, time
from contextlib import redirect_stdout
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zotmer_amd import cli, native, synth
from zotmer_amd.library import engine

def write_fastq(path, R, L, first=0, genome=100_000_000):
    ctx = engine.context()
    with open(path, "wb") as f:
        step = 2_000_000
        for a in range(0, R, step):
            m = min(step, R - a)
            s = ctx.synth_reads(synth.DEFAULT_SEED, first + a, m, L, genome=genome, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.0005)).to_host()
            seq = s.reshape(m, L + 1)
            rec = np.empty((m, 13 + (L + 1) + 2 + (L + 1)), dtype=np.uint8)
            ids = np.char.zfill((np.arange(a, a + m)).astype(str), 10)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
            rec[:, 2:12] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(m, 10)
            rec[:, 12] = ord("\n")
            rec[:, 13:13 + L + 1] = seq
            rec[:, 13 + L + 1] = ord("+"); rec[:, 14 + L + 1] = ord("\n")
            rec[:, 15 + L + 1:15 + 2 * L + 1] = ord("I"); rec[:, 15 + 2 * L + 1] = ord("\n")
            f.write(rec.tobytes())

def run(*argv):
    t0 = time.perf_counter()
    buf = io.StringIO()
    with redirect_stdout(buf):
        cli.main_inner([str(a) for a in argv])
    return time.perf_counter() - t0, buf.getvalue()

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
out = {"reads": R}
t0 = time.perf_counter(); write_fastq("/tmp/a.fastq", R, 150); write_fastq("/tmp/b.fastq", R, 150, first=R)
out["fastq_bytes_each"] = os.path.getsize("/tmp/a.fastq"); out["write_fastq_s"] = time.perf_counter() - t0
os.environ["ZOT_TIMING"] = "1"
for nm in ("a", "b"):
    t, _ = run("kmerize", 25, "/tmp/%s.k25" % nm, "/tmp/%s.fastq" % nm)
    out["kmerize_%s_s" % nm] = t
out["k25_bytes"] = os.path.getsize("/tmp/a.k25")
out["merge_s"], _ = run("merge", "/tmp/m.k25", "/tmp/a.k25", "/tmp/b.k25")
out["dist_s"], txt = run("dist", "-M", "jaccard.qual", 25, "/tmp/a.k25", "/tmp/b.k25")
out["dist_out"] = txt.strip().split("\n")[-1]
out["trim_s"], _ = run("trim", "-c", 3, "/tmp/t.k25", "/tmp/a.k25")
out["instances_per_s_kmerize_e2e"] = 2 * R * 126 / out["kmerize_a_s"]
print(json.dumps(out, indent=1))
