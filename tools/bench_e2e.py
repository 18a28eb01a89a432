#!/usr/bin/env python3
"""End-to-end `zot kmerize` / `merge` / `dist` / `trim` on files (run on the GPU box): where does the wall
time go once the kernels are fast?  Writes a synthetic FASTQ of N reads to /tmp first."""
import io, json, os, re, sys, time
os.environ.setdefault("ZOT_TIMING", "2")          # library/engine.py reads it at import
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
    """-> (seconds, stdout, the [engine] phase lines of ZOT_TIMING=2 summed by phase name, in ms)"""
    import tempfile
    t0 = time.perf_counter()
    buf = io.StringIO()
    err = tempfile.TemporaryFile(mode="w+")
    saved = os.dup(2)
    os.dup2(err.fileno(), 2)
    try:
        with redirect_stdout(buf):
            cli.main_inner([str(a) for a in argv])
    finally:
        sys.stderr.flush()
        os.dup2(saved, 2)
        os.close(saved)
    dt = time.perf_counter() - t0
    err.seek(0)
    phases = {}
    for line in err.read().splitlines():
        m = re.match(r"\s*\[engine\] (.+?)\s+([\d.]+) ms", line)
        if m:
            name = re.sub(r"\d+", "N", m.group(1))          # sizes out of the phase names: one row per kind of phase
            phases[name] = phases.get(name, 0.0) + float(m.group(2))
    return dt, buf.getvalue(), phases

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
only_kmerize = "--kmerize-only" in sys.argv          # config-2 scale: one 15.7 GB FASTQ, the kmerize command alone
out = {"reads": R}
t0 = time.perf_counter(); write_fastq("/tmp/a.fastq", R, 150)
if not only_kmerize:
    write_fastq("/tmp/b.fastq", R, 150, first=R)
out["fastq_bytes_each"] = os.path.getsize("/tmp/a.fastq"); out["write_fastq_s"] = time.perf_counter() - t0
# warm page cache state is what it is after writing the file; the first kmerize also pays for the library's memory (cold),
# the second is the steady state
for rep in ("cold", "warm"):
    for nm in (("a",) if only_kmerize else ("a", "b")):
        if os.path.exists("/tmp/%s.k25" % nm):
            os.remove("/tmp/%s.k25" % nm)          # (truncating a 7 GB file that sits in the page cache costs 0.6 s of its own)
        t, _, ph = run("kmerize", 25, "/tmp/%s.k25" % nm, "/tmp/%s.fastq" % nm)
        out["kmerize_%s_%s_s" % (nm, rep)] = t
        if ph:
            out["kmerize_%s_%s_phases_ms" % (nm, rep)] = {k: round(v, 1) for k, v in ph.items()}
if "--io-sweep" in sys.argv:
    # the write path by its knobs, same process, warm: threads per pwrite, fallocate before the member is written
    sweep = []
    for threads, nofalloc in ((8, 1), (8, 0), (16, 0), (16, 1), (4, 0)):
        native.Context.IO_THREADS = threads
        if nofalloc:
            os.environ["ZOT_NO_FALLOCATE"] = "1"
        else:
            os.environ.pop("ZOT_NO_FALLOCATE", None)
        if os.path.exists("/tmp/a.k25"):
            os.remove("/tmp/a.k25")
        t, _, ph = run("kmerize", 25, "/tmp/a.k25", "/tmp/a.fastq")
        sweep.append({"io_threads": threads, "fallocate": not nofalloc, "kmerize_s": round(t, 3),
                      "device_to_file_ms": round(sum(v for k, v in ph.items() if "file" in k), 1)})
    out["io_sweep"] = sweep
    os.environ.pop("ZOT_NO_FALLOCATE", None)
out["k25_bytes"] = os.path.getsize("/tmp/a.k25")
if not only_kmerize:
    out["merge_s"], _, ph = run("merge", "/tmp/m.k25", "/tmp/a.k25", "/tmp/b.k25")
    out["merge_phases_ms"] = ph
    out["dist_s"], txt, ph = run("dist", "-M", "jaccard.qual", 25, "/tmp/a.k25", "/tmp/b.k25")
    out["dist_out"] = txt.strip().split("\n")[-1]
    out["trim_s"], _, ph = run("trim", "-c", 3, "/tmp/t.k25", "/tmp/a.k25")
out["instances_per_s_kmerize_e2e"] = 2 * R * 126 / out["kmerize_a_warm_s"]
for f in ("a.fastq", "b.fastq", "a.k25", "b.k25", "m.k25", "t.k25"):
    try:
        os.remove("/tmp/" + f)
    except OSError:
        pass
print(json.dumps(out, indent=1))
