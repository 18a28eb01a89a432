#!/usr/bin/env python3
"""
bench.py -- `zot kmerize` K=25 on synthetic 150 bp reads, the metric of BASELINE.json, plus the other single-GPU
configurations as `extra` blocks of the same JSON line.

  python bench.py --gpus N --steps K --warmup W
  (N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Headline.  A step is one whole kmerize of the batch: base stream resident in HBM -> sorted distinct k-mers of both
strands + counts in HBM (hist and acgt included), i.e. zk_kmerize + zk_hist through the C-ABI.  At N = 1 the workload
is BASELINE config 2 (50 M x 150 bp genome-sampled reads, K = 25).  At N > 1 every rank kmerizes its own 50 M reads
(weak scaling) and the per-rank tables are then exchanged by k-mer owner (balanced value ranges by default) with one
RCCL all-to-all and union-summed, so that each rank ends up owning one piece of the global table.

`value` counts emitted k-mer instances (both strands, the unit the reference counts at commands/kmerize.py:523-525) per
second of wall time over the timed steps.  `roofline` is whichever of the two full-size sort kernels takes more of a step
-- the array pass (8 B/key read + the 4-byte tag written) or pass 0 from the base stream (1 B/stream byte + 8 B/key) -- its
algorithmic bytes over its mean launch time by HIP events on the library's own stream, the other and the block dedupe listed
beside it (`others`); `peak` is the 8 TB/s of the spec,
`peak_measured` what a device-to-device copy reaches on this box in this run.  The result of the last timed step is
verified outside the timed region: order-free checksums of the table against the same sums taken straight from the base
stream by an independent encoder, and strict ascent of the k-mers (`verified_checksums`, `verified_ascending`).  `cpu_baseline`: the C oracle on one core on a prefix of the same
reads, and the pure-Python restatement of the reference path on BASELINE config 1.

`extra` (N = 1): config 3 (`zot dist` on two 100 M-k-mer sets), one GPU's share of config 4 (merge of 8 x 50 M-k-mer
sets), one GPU's share of config 5 (K = 31, 37.5 M reads), config 2 again with the base stream starting in pinned
host memory (H2D inside the timed region), and the two inputs the headline does not cover: reads without repeats
(SURVEY 8(d)'s uniform workload) and reads of varying length.  N > 1: `zot merge` (8 sets per GPU, config 4 at N = 8) and `zot dist`
(config 3 sharded over the GPUs) through the product functions of zotmer_amd/parallel.py.  Every block is verified.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy reaches
M64 = (1 << 64) - 1


def model_bytes(n_stream_bytes, instances, unique, K):
    """unfused per-stage byte model the fraction is always quoted against."""
    P = -(-2 * K // 8)
    return n_stream_bytes + 8 * instances + (1 + 2 * P) * 8 * instances + 8 * instances + 12 * unique


def kernel_source_hash():
    """sha256 over the kernel sources: a committed PMC profile is only quoted while the kernels are the profiled ones"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "zotmer_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def measured_traffic(kernel_substr, n_keys_now):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC profile of THIS command
    (profiles/*/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950; tools/collect_traffic.py).  PMC counters cannot be read from inside the
    timed run, so the figure is only quoted when the profile was taken from these very kernel sources (source hash) and
    the same number of keys per launch; otherwise null."""
    prof = os.path.join(ROOT, "profiles")
    best = None
    for sub in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        path = os.path.join(prof, sub, "pmc_traffic.json")
        if os.path.exists(path):
            try:
                d = json.load(open(path))
            except Exception:
                continue
            if d.get("_kernel_source_sha256") != kernel_source_hash():
                continue
            subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
            for name, v in d.items():
                if isinstance(v, dict) and any(x in name for x in subs) and v.get("launches", 0) >= 1 and v.get("traffic_bytes_per_launch", 0) > 1e9 \
                        and abs(v.get("n_keys", 0) - n_keys_now) <= 0.001 * max(n_keys_now, 1):
                    best = dict(bytes=v["traffic_bytes_per_launch"], profile="profiles/%s/pmc_traffic.json" % sub)
    return best


def measured_step_traffic(n_keys_now):
    """HBM bytes of a whole step (all kernels) from the newest committed PMC profile of these kernel sources, or None"""
    prof = os.path.join(ROOT, "profiles")
    best = None
    for sub in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        path = os.path.join(prof, sub, "pmc_traffic.json")
        if os.path.exists(path):
            try:
                d = json.load(open(path))
            except Exception:
                continue
            if d.get("_kernel_source_sha256") != kernel_source_hash() or "_traffic_bytes_per_step" not in d:
                continue
            keys = [v.get("n_keys") for v in d.values() if isinstance(v, dict) and v.get("n_keys")]
            if keys and abs(keys[0] - n_keys_now) <= 0.001 * max(n_keys_now, 1):
                best = dict(bytes=d["_traffic_bytes_per_step"], profile="sum over all kernels of a step, profiles/%s/pmc_traffic.json" % sub)
    return best


def cpu_baseline(cfg, seed, budget_s=12.0):
    """(a) single-core C oracle on a prefix of the same reads, sized to about budget_s seconds; (b) the pure-Python
    restatement of the reference path on BASELINE config 1."""
    from oracle import py_restatement as pr
    from oracle import zkoracle as zo
    from zotmer_amd import synth
    kw = dict(genome=cfg["genome"], sub_thr=synth.frac32(cfg["sub"]), n_thr=synth.frac32(cfg["n"]))
    L, K = cfg["L"], cfg["K"]

    def run(n_reads):
        m = synth.reads_matrix(seed, 0, n_reads, L, **kw)
        offs = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)
        t0 = time.perf_counter()
        r = zo.kmerize_packed(K, m.tobytes(), offs)
        zo.hist(r["counts"])
        return time.perf_counter() - t0, sum(r["acgt"])

    t, inst = run(20000)
    n = int(min(max(20000 * budget_s / max(t, 1e-3), 20000), 2_000_000))
    t, inst = run(n)
    out = {"value": inst / t / 1e9, "unit": "Gk-mers/s", "cores": 1, "kind": "port",
           "sample": "first %d of the %d reads (same generator, same K), C oracle oracle/zk_oracle.c: per-read window loop, "
                     "MSD radix + qsort, RLE merge; %.1f s" % (n, cfg["reads"], t)}
    c1 = synth.CONFIGS["config1"]
    reads = synth.read_strings(seed, 0, c1["reads"], c1["L"], genome=c1["genome"], sub_thr=synth.frac32(c1["sub"]), n_thr=synth.frac32(c1["n"]))
    t0 = time.perf_counter()
    r = pr.kmerize(c1["K"], reads)
    t1 = time.perf_counter() - t0
    out["python_restatement"] = {"value": r["instances"] / t1 / 1e9, "unit": "Gk-mers/s", "cores": 1, "kind": "python-restatement",
                                 "sample": "BASELINE config 1 (%d x %d bp, K = %d) whole, CPython %d.%d, oracle/py_restatement.py: per-read window "
                                           "loop, bucket + list.sort, RLE merge, delta + codec64 (checked against the reference's digests in "
                                           "tests/test_oracle_golden.py); %.1f s" % (c1["reads"], c1["L"], c1["K"], sys.version_info[0],
                                                                                    sys.version_info[1], t1)}
    return out


def timed(ctx, fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    ctx.sync()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    prof = ctx.profile_read()
    ctx.profile(False)
    kern = {n: dict(launches=v["launches"] // steps, ms_per_step=v["ms"] / steps, GBps=(v["bytes"] / 1e9) / (v["ms"] / 1e3) if v["ms"] else None)
            for n, v in prof.items()}
    return dt, r, kern


def copy_peak(ctx, nbytes=4 << 30, reps=5):
    """what a device-to-device copy reaches here (read + written bytes per second), GB/s"""
    a, b = ctx.empty(nbytes, np.uint8), ctx.empty(nbytes, np.uint8)
    best = 0.0
    for _ in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        ctx._check(ctx.lib.zk_copy(ctx.h, b.ptr, a.ptr, nbytes))
        ctx.sync()
        best = max(best, 2 * nbytes / (time.perf_counter() - t0) / 1e9)
    del a, b
    return best


def kernel_table(prof, steps):
    """per-kernel time and algorithmic rate of one step, from the library's HIP-event records"""
    return {n: dict(launches=v["launches"] // steps, ms_per_step=v["ms"] / steps,
                    GBps=(v["bytes"] / 1e9) / (v["ms"] / 1e3) if v["ms"] else None,
                    frac_of_peak=(v["bytes"] / 1e9) / (v["ms"] / 1e3) / HBM_PEAK_GBS if v["ms"] else None)
            for n, v in prof.items()}


def extra_kmerize_variant(ctx, name, stream, K, steps, note, cap):
    """zk_kmerize + zk_hist on another kind of input, verified the same way as the headline (cap: entries the table may have)"""
    out_k, out_c = ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32)

    def step():
        k, c, st = ctx.kmerize(stream, K, 0, out=(out_k, out_c))
        return k, c, st, ctx.hist(c)

    step()
    ctx.sync()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        k, c, st, h = step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    prof = ctx.profile_read()
    ctx.profile(False)
    ok = ctx.checksum(k, c) == ctx.stream_checksum(stream, K)
    asc = ctx.first_descent(k) == k.n
    return {"workload": note, "value": st.n_instances / dt / 1e9, "unit": "Gk-mers/s", "ms_per_step": dt * 1e3,
            "instances_per_step": st.n_instances, "unique": st.n_unique, "canonical_unique": st.n_canonical,
            "verified": bool(ok and asc), "verified_by": "order-free checksums against the independent stream encoder + strict ascent of the k-mers",
            "kernels": kernel_table(prof, steps)}


def add_sums(a, b):
    return tuple((x + y) & M64 for x, y in zip(a, b))


# ---- extra blocks (N = 1) -------------------------------------------------------------------------------------
def extra_config3(ctx, steps, scale):
    """BASELINE config 3: `zot dist` on two sorted sets of 100 M 50-bit k-mers, half shared.  Timed: Measure.prep on both
    files as the command does it (library/engine.py measure_prep: at K = fK a read-only check of the strict ascent, no copy)
    + dist.split (zk_split).
    Verified: (a) the generator's construction (two windows of one key sequence overlap in exactly half), (b) an
    independent path -- concatenate, radix sort, run-length count: the number of runs of length 2 is |X & Y|."""
    from zotmer_amd import synth
    c3 = synth.CONFIG3
    n = int(c3["n"] * scale)
    ka, _ = ctx.synth_set(c3["seed"], 0, n, c3["key_bits"], counts=False)
    kb, _ = ctx.synth_set(c3["seed"], n // 2, n, c3["key_bits"], counts=False)
    from zotmer_amd.library import engine

    def prep_and_split():          # what the command runs per pair: commands/dist.py prep() on both files, then the split
        return ctx.split(engine.measure_prep(ctx, ka, 0), engine.measure_prep(ctx, kb, 0))

    dt_all, abc, kern_all = timed(ctx, prep_and_split, steps)
    dt, abc2, kern = timed(ctx, lambda: ctx.split(ka, kb), steps)
    # independent check of a
    cat = ctx.empty(ka.n + kb.n, np.uint64)
    ctx._check(ctx.lib.zk_copy(ctx.h, cat.ptr, ka.ptr, ka.nbytes))
    ctx._check(ctx.lib.zk_copy(ctx.h, cat.ptr + ka.nbytes, kb.ptr, kb.nbytes))
    ctx.sync()
    u, cnt = ctx.sort_count(cat, c3["key_bits"])
    h = ctx.hist(cnt)
    ok = abc == abc2 and h.get(2, 0) == abc[0] and set(h) <= {1, 2} and abc[0] + abc[1] == ka.n and abc[0] + abc[2] == kb.n \
        and u.n == abc[0] + abc[1] + abc[2]
    byts = 8 * (ka.n + kb.n)
    ik = kern.get("intersect", {})
    return {"workload": "BASELINE config 3: zot dist (jaccard) on two sorted sets of %d and %d 50-bit k-mers, %d shared" % (ka.n, kb.n, abc[0]),
            "value": (ka.n + kb.n) / dt / 1e9, "unit": "Gk-mers/s", "ms_per_step": dt * 1e3, "ms_with_prep": dt_all * 1e3,
            "abc": list(abc), "jaccard_distance": (abc[1] + abc[2]) / float(sum(abc)),
            "verified": bool(ok), "verified_by": "sort + run-length count of the concatenation (hist[2] == a), set sizes",
            "roofline": {"bound": "hbm", "kernel": "intersect_kernel (merge-path intersect count)", "algorithmic_bytes": byts,
                         "achieved": (byts / 1e9) / (ik["ms_per_step"] / 1e3) if ik.get("ms_per_step") else None,
                         "achieved_wall": byts / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (byts / 1e9) / (ik["ms_per_step"] / 1e3) / HBM_PEAK_GBS if ik.get("ms_per_step") else None},
            "kernels": kern_all}


def extra_trim(ctx, steps, scale):
    """`zot trim` (commands/trim.py:54-62) at the size of a config-4 set: 50 M (k-mer, 64-bit count) pairs with geometric counts (mean 8),
    entries with a count in [4, inf) kept.  Verified: the number kept is the input histogram's tail, every count kept is >= 4, the
    k-mers still ascend strictly, and the output's count histogram is the input's from 4 on."""
    from zotmer_amd import synth
    a = synth.config4_set_args(0, scale)
    k, c = ctx.synth_set(a["seed"], a["first"], a["count"], a["key_bits"], mul=a["mul"], add=a["add"], mod=a["mod"])
    lo = 4
    ok_, oc_ = ctx.empty(k.n, np.uint64), ctx.empty(k.n, np.uint64)
    n = C.c_uint64(0)

    def run():
        ctx._check(ctx.lib.zk_trim(ctx.h, k.ptr, c.ptr, 64, k.n, lo, 0, ok_.ptr, oc_.ptr, k.n, C.byref(n)))
        return n.value

    dt, kept, kern = timed(ctx, run, steps)
    hin = ctx.hist(c)
    tk, tc = ok_.view(kept), oc_.view(kept)
    hout = ctx.hist(tc) if kept else {}
    ok = kept == sum(v for x, v in hin.items() if x >= lo) and hout == {x: v for x, v in hin.items() if x >= lo} and ctx.first_descent(tk) == kept
    byts = 16 * k.n + 16 * kept
    sk = kern.get("select", {})
    return {"workload": "zot trim of a set of %d (k-mer, 64-bit count) pairs (geometric counts, mean 8): counts >= %d kept" % (k.n, lo),
            "value": k.n / dt / 1e9, "unit": "G (k-mer,count) pairs/s", "ms_per_step": dt * 1e3, "pairs_in": k.n, "pairs_out": kept,
            "verified": bool(ok), "verified_by": "the input histogram's tail == the number kept == the output's histogram; strict ascent",
            "roofline": {"bound": "hbm", "kernel": "select_kernel<TrimOp> (flag, decoupled look-back, compact)", "algorithmic_bytes": byts,
                         "achieved": (byts / 1e9) / (sk["ms_per_step"] / 1e3) if sk.get("ms_per_step") else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (byts / 1e9) / (sk["ms_per_step"] / 1e3) / HBM_PEAK_GBS if sk.get("ms_per_step") else None},
            "kernels": kern}


def extra_config4_share(ctx, steps, scale):
    """One GPU's share of BASELINE config 4: the k-way merge (mergeNinto) of 8 sets of 50 M k-mers with geometric 64-bit
    counts drawn from the shared 200 M-key pool.  Verified by the checksum of checksums (sum over the inputs of
    (count, key*count, murmer(key)*count) == the same sums over the merged set) and strict sortedness via the run-length
    count of the output keys (every run has length 1)."""
    from zotmer_amd import synth
    sets, sums, total = [], (0, 0, 0), 0
    for s in range(8):
        a = synth.config4_set_args(s, scale)
        k, c = ctx.synth_set(a["seed"], a["first"], a["count"], a["key_bits"], mul=a["mul"], add=a["add"], mod=a["mod"])
        sets.append((k, c))
        sums = add_sums(sums, ctx.checksum_counts(k, c))
        total += k.n
    ok_, oc_ = ctx.empty(total, np.uint64), ctx.empty(total, np.uint64)
    dt, (mk, mc, acgt), kern = timed(ctx, lambda: ctx.merge_n(sets, out=(ok_, oc_)), steps)
    got = ctx.checksum_counts(mk, mc)
    chk = ctx.copy_of(mk)
    _, runs = ctx.rle(chk)
    hr = ctx.hist(runs)
    ok = got == sums and hr == {1: mk.n} and sum(acgt) == sums[0]
    # one pass over the data (kway.hip: every pair read once, the union written once), or -- ZOT_TUNE=kway=0 -- three levels of 2-way
    # passes that read every pair once each (upper bound on the reads) + the output
    one_pass = "kway=0" not in os.environ.get("ZOT_TUNE", "")
    byts = (16 * total + 16 * mk.n) if one_pass else (3 * 16 * total + 16 * mk.n)
    return {"workload": "one GPU's share of BASELINE config 4: zot merge of 8 sets x %d k-mers (pool %d, geometric counts), 64-bit counts"
                        % (sets[0][0].n, synth.config4_set_args(0, scale)["mod"]),
            "value": total / dt / 1e9, "unit": "G (k-mer,count) pairs/s", "ms_per_step": dt * 1e3, "pairs_in": total, "unique_out": mk.n,
            "verified": bool(ok), "verified_by": "checksum of checksums + run-length count of the output (strictly ascending)",
            "roofline": {"bound": "hbm", "kernel": "kway_merge_kernel<u64, 8> (one pass: sampled splitters, tiles merged in LDS)" if one_pass else "union_sum_kernel<u64> x 3 tree levels",
                         "algorithmic_bytes": byts,
                         "achieved": byts / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byts / dt / 1e9 / HBM_PEAK_GBS},
            "kernels": kern}


def extra_config5_share(ctx, scale, synth, seed, batches=4):
    """One GPU's share of BASELINE config 5 (K = 31, 300 M reads / 8 = 37.5 M x 150 bp from a 3.1 Gbp genome): at ~1.8x coverage
    nearly every k-mer is distinct (about 5.6 G entries = 68 GB for one GPU), so the share is counted the way `zot kmerize`
    does it -- library/engine.py KmerTable: batches, each sorted and counted straight into the table slab, tables
    union-summed pairwise.  Timed: from the resident base stream of each batch to the final table + hist.  The procedure
    runs twice: `cold_ms` includes growing the table slab and the sort workspace (hipMalloc costs ~25 ms per GB on this
    stack and is paid once per process), `value` is the second run with the memory in place.  The synthetic streams are
    generated outside the timed region."""
    from zotmer_amd.library import engine
    c5 = synth.CONFIGS["config5"]
    R5, L, K = int(c5["reads"] // 8 * scale), c5["L"], c5["K"]
    per = -(-R5 // batches)
    ctx.release_workspace()                      # config 2 left a 120 GB sort arena behind
    runs = []
    for attempt in range(2):
        table = engine.KmerTable(ctx, K)
        table.expect(R5 * (L + 1))
        want, t_total = (0, 0, 0), 0.0
        for b in range(batches):
            n = min(per, R5 - b * per)
            if n <= 0:
                break
            d = ctx.synth_reads(seed, b * per, n, L, genome=c5["genome"], sub_thr=synth.frac32(c5["sub"]), n_thr=synth.frac32(c5["n"]))
            if attempt == 0:
                want = add_sums(want, ctx.stream_checksum(d, K))
            ctx.sync()
            t0 = time.perf_counter()
            table.add_device_stream(d)
            ctx.sync()
            t_total += time.perf_counter() - t0
            del d
        t0 = time.perf_counter()
        k, c, h = table.device_result()
        ctx.sync()
        t_total += time.perf_counter() - t0
        runs.append(t_total)
        if attempt == 0:
            want0 = want
    inst = table.instances
    ok = ctx.checksum(k, c) == want0 and inst == want0[0] and ctx.first_descent(k) == k.n
    n_unique = k.n
    mb = model_bytes(R5 * (L + 1), inst, n_unique, K)
    slab_bytes = 12 * (table.slab.E + table.scratch.E)
    del k, c, table
    engine.release_table_memory(ctx)
    ctx.release_workspace()
    t = runs[1]
    return {"workload": "one GPU's share of BASELINE config 5: zot kmerize k=%d, %d x %d bp reads (300 M / 8), genome %d, in %d batches "
                        "(library/engine.py KmerTable: per-batch sort + count into the table slab, pairwise union-sum), hist included"
                        % (K, R5, L, c5["genome"], batches),
            "value": inst / t / 1e9, "unit": "Gk-mers/s", "ms_total": t * 1e3, "cold_ms": runs[0] * 1e3, "table_slab_bytes": slab_bytes,
            "instances": inst, "unique": n_unique,
            "verified": bool(ok), "verified_by": "order-free checksums of the final table == the sums taken from the base streams of all batches; "
                                                 "k-mers strictly ascending",
            "roofline": {"bound": "hbm", "model_bytes": mb, "speedup_over_unfused_model_at_peak": mb / t / 1e9 / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s"}}


def extra_uniform_reads(ctx, sc, synth, seed, cfg, K, L):
    """SURVEY 8(d)'s secondary workload, "worst case U ~ I": iid bases, no k-mer occurs twice -- the look before the sort declines the
    top-bits-first plan, and the keys of BOTH strands are sorted instead: three passes over the top 27 bits, the rest tile by tile in LDS
    (csrc/tilesort.hip), counted there.  Two figures: 20 M reads as ONE batch of zk_kmerize, and 40 M reads the way `zot kmerize` counts an input
    that does not fit one batch -- library/engine.py KmerTable: batches counted as canonical lists, union-summed pairwise, the strands
    rebuilt once.  (50 M such reads do not fit ONE card in any order of work: their table alone is 12.6 G entries = 151 GB, and the
    strands are rebuilt from a 75 GB canonical list through two 50 GB word buffers: 277 GB + the sort arena; 40 M reads: 221 GB.)"""
    from zotmer_amd.library import engine
    Ru = int(20_000_000 * sc)
    uni = ctx.synth_reads(seed + 1, 0, Ru, L, genome=0, sub_thr=0, n_thr=synth.frac32(cfg["n"]))
    out = extra_kmerize_variant(
        ctx, "uniform", uni, K, 2, "SURVEY 8(d) uniform workload: zot kmerize k=%d on %d x %d bp reads of iid bases (no k-mer occurs twice), "
        "one batch: the keys of both strands sorted at once (passes over the top bits + a tile sort that counts)" % (K, Ru, L), 2 * Ru * (L - K + 1) + 1024)
    del uni
    ctx.release_workspace()
    R50, batches = int(40_000_000 * sc), 4
    per = -(-R50 // batches)
    runs, want0 = [], None
    for attempt in range(2):
        table = engine.KmerTable(ctx, K)
        table.expect(R50 * (L + 1))
        want, t_total = (0, 0, 0), 0.0
        for b in range(batches):
            n = min(per, R50 - b * per)
            if n <= 0:
                break
            d = ctx.synth_reads(seed + 1, b * per, n, L, genome=0, sub_thr=0, n_thr=synth.frac32(cfg["n"]))
            if attempt == 0:
                want = add_sums(want, ctx.stream_checksum(d, K))
            ctx.sync()
            t0 = time.perf_counter()
            table.add_device_stream(d)
            ctx.sync()
            t_total += time.perf_counter() - t0
            del d
        t0 = time.perf_counter()
        k, c, h = table.device_result()
        ctx.sync()
        t_total += time.perf_counter() - t0
        runs.append(t_total)
        if attempt == 0:
            want0 = want
    inst = table.instances
    ok = ctx.checksum(k, c) == want0 and inst == want0[0] and ctx.first_descent(k) == k.n
    out["reads_40M_in_batches"] = {
        "workload": "%d x %d bp reads of iid bases in %d batches through library/engine.py KmerTable (canonical lists, pairwise union-sum, strands "
                    "rebuilt once), hist included" % (R50, L, batches),
        "value": inst / runs[1] / 1e9, "unit": "Gk-mers/s", "ms_total": runs[1] * 1e3, "cold_ms": runs[0] * 1e3, "instances": inst, "unique": k.n,
        "verified": bool(ok)}
    del k, c, table
    engine.release_table_memory(ctx)
    ctx.release_workspace()
    return out


def extra_e2e_h2d(ctx, stream, K, steps, out_k, out_c, overlapped=False):
    """config 2 with the base stream starting in PINNED HOST memory -- the H2D copy is inside the timed
    region (sequential: the sort needs the whole batch), the result stays in HBM."""
    host = ctx.pinned(stream.n)
    ctx._check(ctx.lib.zk_download(ctx.h, host.ptr, stream.ptr, stream.n))
    dev = ctx.empty(stream.n, np.uint8)

    def step():
        ctx.upload_async(dev, host.ptr, stream.n)
        k, c, st = ctx.kmerize(dev, K, 0, out=(out_k, out_c))
        return k, c, st, ctx.hist(c)

    t0 = time.perf_counter()
    ctx.upload_async(dev, host.ptr, stream.n)
    ctx.sync()
    t_copy = time.perf_counter() - t0
    dt, (k, c, st, h), kern = timed(ctx, step, steps)
    ok = ctx.checksum(k, c) == ctx.stream_checksum(stream, K)
    out = {"workload": "BASELINE config 2 end to end from pinned host memory: H2D of the %.2f GB base stream + kmerize + hist, result left in HBM"
                       % (stream.n / 1e9), "value": st.n_instances / dt / 1e9, "unit": "Gk-mers/s", "ms_per_step": dt * 1e3,
           "h2d_ms": t_copy * 1e3, "h2d_GBps": stream.n / t_copy / 1e9, "verified": bool(ok)}
    want = ctx.stream_checksum(stream, K)
    del dev, k, c
    if overlapped:
        # (measured, profiles/r04/e2e_h2d_overlap.json: 8 batches cost 134 ms of counting + 40 of merges + 30 of strands where one
        # batch costs 96 in all -- what the copy hides is less than what the batches add; not part of the default line)
        try:
            out["overlapped"] = e2e_h2d_overlapped(ctx, host, stream.n, K, steps, want)
        except Exception as e:          # noqa: BLE001
            out["overlapped"] = {"error": repr(e)}
    host.free()
    return out


def e2e_h2d_overlapped(ctx, host, n_bytes, K, steps, want, batches=8):
    """The same input the way `zot kmerize` takes a file (library/engine.py count_fastq_file + KmerTable): in batches, the copy of batch
    b + 1 running -- on a stream of its own: a second context of the same device -- while batch b is counted and the tables of
    earlier batches are union-summed; the strands are rebuilt once at the end.  What cannot hide behind the copy is the last batch,
    the top of the merge tree and the strands."""
    from zotmer_amd import native
    from zotmer_amd.library import engine
    rec = int(np.frombuffer((C_char_at(host.ptr, 4096)), dtype=np.uint8).tolist().index(10)) + 1          # bytes per record (uniform reads)
    recs = n_bytes // rec
    per = -(-recs // batches) * rec
    copy_ctx = native.Context(ctx.device if hasattr(ctx, "device") else 0)
    bufs = [ctx.empty(per, np.uint8), ctx.empty(per, np.uint8)]
    cuts = [(b * per, min(per, n_bytes - b * per)) for b in range(batches) if b * per < n_bytes]
    engine.release_table_memory(ctx)

    def run():
        table = engine.KmerTable(ctx, K)
        table.expect(n_bytes)
        copy_ctx.upload_async(bufs[0], host.ptr + cuts[0][0], cuts[0][1])
        for b, (off, nb) in enumerate(cuts):
            copy_ctx.sync()                                   # batch b has arrived
            if b + 1 < len(cuts):
                ctx.sync()                                    # (the buffer batch b + 1 goes into was counted two batches ago)
                copy_ctx.upload_async(bufs[(b + 1) & 1], host.ptr + cuts[b + 1][0], cuts[b + 1][1])
            table.add_device_stream(bufs[b & 1].view(nb))
        k, c, h = table.device_result()
        ctx.sync()
        return table, k, c, h

    run()
    run()          # (the second run still grows the table memory once: the slabs change roles)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        table, k, c, h = run()
    dt = (time.perf_counter() - t0) / steps
    ok = ctx.checksum(k, c) == want and ctx.first_descent(k) == k.n
    inst = table.instances
    del k, c, table, bufs
    copy_ctx.close()
    engine.release_table_memory(ctx)
    return {"workload": "the same, as `zot kmerize` takes a file: %d batches, the copy of the next batch (a copy stream of its own) behind the counting "
                        "of the current one, tables union-summed pairwise, strands rebuilt once" % len(cuts),
            "value": inst / dt / 1e9, "unit": "Gk-mers/s", "ms_per_step": dt * 1e3, "verified": bool(ok)}


def C_char_at(ptr, n):
    import ctypes
    return ctypes.string_at(ptr, n)


def extra_e2e_h2d_only(ctx, synth, seed, cfg, K, L, sc):
    R = int(cfg["reads"] * sc)
    stream = ctx.synth_reads(seed, 0, R, L, genome=cfg["genome"], sub_thr=synth.frac32(cfg["sub"]), n_thr=synth.frac32(cfg["n"]))
    est_unique = int(2 * (min(cfg["genome"], R * L) + R * L * cfg["sub"] * 22) * 1.25) + (1 << 20)
    cap = min(est_unique, 2 * stream.n)
    out_k, out_c = ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32)
    return extra_e2e_h2d(ctx, stream, K, 2, out_k, out_c, overlapped=True)


# ---- multi-GPU extras ------------------------------------------------------------------------------------------
def extra_merge_multi(ctx, ex, steps, scale, world, rank):
    """`zot merge` over the GPUs through parallel.Exchange.merge_sets: every rank merges its own 8 sets (sets rank, rank + world,
    ... of 8 * world; BASELINE config 4 at world = 8), one exchange by k-mer owner, merge of the received pieces."""
    from zotmer_amd import synth
    sets, sums, total = [], (0, 0, 0), 0
    for j in range(8):
        a = synth.config4_set_args(rank + j * world, scale)
        k, c = ctx.synth_set(a["seed"], a["first"], a["count"], a["key_bits"], mul=a["mul"], add=a["add"], mod=a["mod"])
        sets.append((k, c))
        sums = add_sums(sums, ctx.checksum_counts(k, c))
        total += k.n
    lk, lc = ctx.empty(total, np.uint64), ctx.empty(total, np.uint64)

    def step():
        k, c, _ = ctx.merge_n(sets, out=(lk, lc))
        kt, ct, n = ex.ops.to_tensors(k, c)
        return ex.merge_sets(kt, ct, n)

    for _ in range(1):
        step()
    ex.comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    ctx.sync()
    ex.comm.barrier()
    dt = (time.perf_counter() - t0) / steps
    dt = ex.comm.all_reduce([int(dt * 1e9)], "max")[0] / 1e9
    ok = ex.verify_global(res["k"], res["c"], sums)
    tot_all = ex.comm.all_reduce([total])[0]
    owned = ex.comm.all_gather_object(res["k"].n)
    return {"workload": "zot merge of %d sets x %d k-mers over %d GPUs (8 sets per GPU; BASELINE config 4 at 8 GPUs), owner = %s, transport = %s"
                        % (8 * world, sets[0][0].n, world, ex.owner, ex.comm.name),
            "value": tot_all / dt / 1e9, "unit": "G (k-mer,count) pairs/s", "ms_per_step": dt * 1e3, "pairs_in": tot_all,
            "unique_out": res["n_global"], "owned_per_rank": owned, "balance_max_over_mean": max(owned) / (sum(owned) / float(world)),
            "scaling": "weak", "verified": bool(ok), "verified_by": "checksum of checksums all-reduced over the ranks"}


def extra_dist_multi(ctx, ex, steps, scale, world, rank):
    """`zot dist` on BASELINE config 3 sharded over the GPUs (strong scaling: the two 100 M-k-mer sets are fixed): every rank
    holds the rank-th contiguous piece of each sorted set, one exchange by owner, zk_split, all-reduce of (a, b, c)."""
    import torch
    from zotmer_amd import synth
    c3 = synth.CONFIG3
    n = int(c3["n"] * scale)
    ka, _ = ctx.synth_set(c3["seed"], 0, n, c3["key_bits"], counts=False)
    kb, _ = ctx.synth_set(c3["seed"], n // 2, n, c3["key_bits"], counts=False)
    want = ctx.split(ka, kb)

    def piece(k):
        lo, hi = k.n * rank // world, k.n * (rank + 1) // world
        t = torch.empty(max(hi - lo, 1), dtype=torch.int64, device="cuda")
        if hi > lo:
            ctx._check(ctx.lib.zk_copy(ctx.h, t.data_ptr(), k.ptr + 8 * lo, 8 * (hi - lo)))
        ctx.sync()
        return t, hi - lo

    (xt, nx), (yt, ny) = piece(ka), piece(kb)
    na, nb = ka.n, kb.n
    del ka, kb
    ex.dist_pair(xt, nx, yt, ny)
    ex.comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        abc, sizes = ex.dist_pair(xt, nx, yt, ny)
    ctx.sync()
    ex.comm.barrier()
    dt = (time.perf_counter() - t0) / steps
    dt = ex.comm.all_reduce([int(dt * 1e9)], "max")[0] / 1e9
    return {"workload": "zot dist on BASELINE config 3 (%d and %d k-mers) sharded over %d GPUs, owner = %s, transport = %s"
                        % (na, nb, world, ex.owner, ex.comm.name),
            "value": (na + nb) / dt / 1e9, "unit": "Gk-mers/s", "ms_per_step": dt * 1e3, "abc": list(abc), "scaling": "strong",
            "verified": bool(tuple(abc) == tuple(want) and tuple(sizes) == (na, nb)),
            "verified_by": "(a, b, c) equals the single-GPU zk_split of the whole sets"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: config 2 = 50 M)")
    ap.add_argument("--both", action="store_true", help="sort both strands literally instead of canonical + mirror")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the checksum verification of the timed result")
    ap.add_argument("--verify", action="store_true", help="(default; kept for round-1 command lines)")
    ap.add_argument("--no-extras", action="store_true", help="headline only")
    ap.add_argument("--extras-scale", type=float, default=1.0, help="shrink the extra workloads (tests)")
    ap.add_argument("--only-extra", default="", help="run just this extra block (development): skips the headline")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_exchange = os.environ.get("ZOT_FORCE_EXCHANGE") == "1"      # rehearse the N > 1 path with one rank
    if force_exchange and world == 1:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
    if world > 1 or force_exchange:
        # torch bundles its own HIP runtime: let it load first so that libzotk.so binds to the same
        # copy (same SONAME) instead of dragging a second runtime into the process
        import torch  # noqa: F401
    # a rebuild (make) or RCCL's version banner may write to stdout; the contract is ONE JSON line there,
    # so everything but the final print goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import __graft_entry__ as ge
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")

    dist = None
    if world > 1 or force_exchange:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    # one rank builds (normally a no-op: the library travels prebuilt), the others wait for it before they dlopen
    if rank == 0:
        ge.build()
    if dist is not None and world > 1:
        dist.barrier()
    from zotmer_amd import native, synth

    cfg = dict(synth.CONFIGS["config2"])
    if a.reads:
        cfg["reads"] = a.reads
    R, L, K = cfg["reads"], cfg["L"], cfg["K"]
    seed = synth.DEFAULT_SEED
    flags = native.KMERIZE_BOTH if a.both else native.KMERIZE_CANONICAL

    ctx = native.Context(local)
    if os.environ.get("ZOT_TUNE"):          # A/B runs of the library's knobs, e.g. ZOT_TUNE=wide_tiles=0,early_collapse=2 (recorded in the line)
        ctx.tune(**{k: int(v) for k, v in (kv.split("=") for kv in os.environ["ZOT_TUNE"].split(","))})
    if a.only_extra:
        fn = {"config3_dist": lambda: extra_config3(ctx, 5, a.extras_scale),
              "trim": lambda: extra_trim(ctx, 5, a.extras_scale),
              "config4_merge_share": lambda: extra_config4_share(ctx, 3, a.extras_scale),
              "config5_share_k31": lambda: extra_config5_share(ctx, a.extras_scale, synth, seed),
              "uniform_reads": lambda: extra_uniform_reads(ctx, a.extras_scale, synth, seed, cfg, K, L),
              "config2_e2e_h2d": lambda: extra_e2e_h2d_only(ctx, synth, seed, cfg, K, L, a.extras_scale)}[a.only_extra]
        r = fn()
        os.dup2(real_stdout, 1)
        print(json.dumps({a.only_extra: r}), flush=True)
        ctx.close()
        return
    n_bytes = R * (L + 1)
    stream = ctx.synth_reads(seed, rank * R, R, L, genome=cfg["genome"], sub_thr=synth.frac32(cfg["sub"]),
                             n_thr=synth.frac32(cfg["n"]))
    ctx.sync()
    # distinct k-mers: 2 strands x (genome + ~21 novel windows per substitution); leave headroom
    est_unique = int(2 * (min(cfg["genome"], R * L) + R * L * cfg["sub"] * 22) * 1.25) + (1 << 20)
    cap = min(est_unique, 2 * n_bytes)
    par, comm_note = None, None
    if dist is not None:
        # the exchange moves torch-owned tensors that the library writes through their data_ptr()
        # (uint64 carried as int64, uint32 as int32)
        import torch
        from zotmer_amd import parallel
        kt = torch.empty(cap, dtype=torch.int64, device="cuda")
        ct = torch.empty(cap, dtype=torch.int32, device="cuda")
        out_k = native.DeviceArray.borrow(ctx, kt.data_ptr(), np.uint64, cap, keep=kt)
        out_c = native.DeviceArray.borrow(ctx, ct.data_ptr(), np.uint32, cap, keep=ct)
        notes = []
        comm = parallel.make_comm(ctx, dist, notes)     # probes the native transport on every rank; a switch is recorded in the line
        comm_note = notes[0] if notes else None
        par = parallel.Exchange(ctx, dist, K, owner=os.environ.get("ZOT_OWNER", "range"), comm=comm)
    else:
        out_k, out_c = ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32)

    def step():
        if par is not None and not a.both:
            # the counted canonical lists are exchanged (half the size of the both-strand tables) and each rank rebuilds the
            # strands of the k-mers whose canonical form it owns: parallel.Exchange.kmerize_finish
            k, c, st = ctx.kmerize(stream, K, flags | native.KMERIZE_CANONICAL_ONLY, out=(out_k, out_c))
            k, c = par.kmerize_finish(kt, ct, k.n)
        else:
            k, c, st = ctx.kmerize(stream, K, flags, out=(out_k, out_c))
            if par is not None:
                if par.owner == "range":
                    par.balanced_cuts([(kt, k.n)])
                k, c = par.exchange_and_merge(kt, ct, k.n)
        h = ctx.hist(c)
        return k, c, st, h

    def fence():
        ctx.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(a.warmup):
        step()
    fence()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        k, c, st, h = step()
    fence()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile(False)

    owned = None
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        total_instances = par.comm.all_reduce([st.n_instances])[0]
        owned = par.comm.all_gather_object(k.n)
    else:
        total_instances = st.n_instances

    verify, ascending = None, None
    if not a.no_verify:
        want = ctx.stream_checksum(stream, K)
        if par is None:
            verify = bool(ctx.checksum(k, c) == want)
        else:
            verify = bool(par.verify_global(k, c, want))
        # the checksums are order-free: the order is checked apart (every rank's piece strictly ascending)
        ascending = bool(ctx.first_descent(k) == k.n)
        if par is not None:
            ascending = bool(par.comm.all_reduce([0 if ascending else 1])[0] == 0)
    peak_measured = copy_peak(ctx) if rank == 0 and not a.no_verify else None          # (not in profiled runs)

    extras = {}
    if not a.no_extras:
        sc = a.extras_scale
        if par is None:
            del k, c
            for name, fn in (("config3_dist", lambda: extra_config3(ctx, 5, sc)),
                             ("trim", lambda: extra_trim(ctx, 5, sc)),
                             ("config4_merge_share", lambda: extra_config4_share(ctx, 3, sc)),
                             ("config2_e2e_h2d", lambda: extra_e2e_h2d(ctx, stream, K, 2, out_k, out_c))):
                try:
                    extras[name] = fn()
                except Exception as e:      # an extra never takes the headline down; the failure is in the line
                    extras[name] = {"error": repr(e)}
            del out_k, out_c
            out_k = out_c = None
            # the inputs the headline does not cover
            try:
                # reads of varying length: every record of config 2 cut in two at a place between its bases 100 and 149
                # (a newline replaces that base), so no two neighbouring records have the same length and the tiles of the
                # first pass follow positions, not records
                R2 = int(R * sc)
                host = stream.to_host(R2 * (L + 1))
                i = np.arange(R2, dtype=np.uint64)
                host[(i * np.uint64(L + 1) + np.uint64(100) + (synth.mix64(i + np.uint64(seed)) % np.uint64(50))).astype(np.int64)] = 10
                var = ctx.upload(host)
                del host, i
                extras["config2_variable_length"] = extra_kmerize_variant(
                    ctx, "var", var, K, 2, "BASELINE config 2 with every record cut in two at a varying place (%d records of 0 .. 149 bases, "
                    "%d stream bytes): tiles of positions, same pipeline" % (2 * R2, var.n), cap)
                del var
            except Exception as e:
                extras["config2_variable_length"] = {"error": repr(e)}
            del stream
            stream = None
            ctx.release_workspace()
            try:
                extras["uniform_reads"] = extra_uniform_reads(ctx, sc, synth, seed, cfg, K, L)
            except Exception as e:
                extras["uniform_reads"] = {"error": repr(e)}
            ctx.release_workspace()
            try:
                extras["config5_share_k31"] = extra_config5_share(ctx, sc, synth, seed)
            except Exception as e:
                extras["config5_share_k31"] = {"error": repr(e)}
        else:
            for name, fn in (("merge_multi_gpu", lambda: extra_merge_multi(ctx, par, 3, sc, par.world, par.rank)),
                             ("dist_multi_gpu", lambda: extra_dist_multi(ctx, par, 5, sc, par.world, par.rank))):
                try:
                    extras[name] = fn()
                except Exception as e:
                    extras[name] = {"error": repr(e)}

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = total_instances * a.steps / dt / 1e9
        # the dominant kernel: whichever of the two full-size sort kernels takes more of a step (since the block dedupe the
        # array pass runs once per step and the pass that reads the base stream is its equal); the other is listed beside it
        cands = {"pass_keys": ("pass_pipe_kernel<array> (the LSD radix pass over the key array, persistent two-stage pipeline: 8 B/key read, and "
                               "written 4 B/key -- the 32-bit tags the block dedupe needs -- or 8 B/key)",
                               # (the tag pass of the default plan: 16 K-key tiles, places from LDS adds; or its other forms by zk_tune)
                               ("pass_pipe_kernel<zk::Cfg<1024, 16, 9, 1, 4, 32, true>, 0, 3>", "pass_pipe_kernel<zk::Cfg<1024, 16, 9, 1, 4, 32, true>, 0, 2>",
                                "pass_pipe_kernel<zk::Cfg<512, 16, 9, 1, 4, 32, true>, 0, 3>", "pass_pipe_kernel<zk::Cfg<512, 16, 9, 1, 4, 32, true>, 0, 2>")),
                 "pass_stream": ("stream_pass0_kernel (pass 0 over static stream ranges: 2-bit image -> canonical 64-bit keys, grouped by digit in LDS, "
                                 "whole 64-byte units out; 1 B/stream byte + 8 B/key)",
                                 "stream_pass0_kernel<9, 8, true, true")}
        empty = dict(launches=0, ms=0.0, bytes=0)
        dom = max(cands, key=lambda n: prof.get(n, empty)["ms"])
        pk = prof.get(dom, empty)
        ach = (pk["bytes"] / 1e9) / (pk["ms"] / 1e3) if pk["ms"] else 0.0
        mb = model_bytes(n_bytes, st.n_instances, st.n_unique, K)
        traffic = measured_traffic(cands[dom][1], st.n_windows)
        step_traffic = measured_step_traffic(st.n_windows)
        others = []
        # the block dedupe beside them (third by time; its bytes: a 4-byte tag read per key + one 8-byte word written per distinct key)
        cands_all = dict(cands, rle=("dedupe2_kernel (block dedupe: an LDS hash table per block of equal top 18 bits counts the copies and leaves "
                                     "the block sorted; two workgroups per CU; 4 B/key read + 8 B/distinct key written)", "dedupe2_kernel"))
        for n in cands_all:
            v = prof.get(n, empty)
            if n != dom and v["ms"]:
                g = (v["bytes"] / 1e9) / (v["ms"] / 1e3)
                o = {"kernel": cands_all[n][0], "achieved": g, "frac": g / HBM_PEAK_GBS, "launches": v["launches"],
                     "avg_launch_ms": v["ms"] / v["launches"]}
                tr = measured_traffic(cands_all[n][1], st.n_windows)
                o["traffic"] = tr["bytes"] if tr else None
                if n == "rle":          # (the tag also times the 0.1 ms run-length count of the look before the sort: per step, not per launch)
                    o["ms_per_step"] = v["ms"] / a.steps
                    del o["avg_launch_ms"]
                others.append(o)
        out = {
            "metric": "Gk-mers/sec kmerize k=25 on synthetic 150bp FASTQ; achieved HBM GB/s fraction",
            "value": value, "unit": "Gk-mers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "zot kmerize k=%d, %d x %dbp genome-sampled synthetic reads per GPU (BASELINE config 2), "
                                   "base stream resident in HBM -> sorted distinct both-strand k-mers + counts + hist"
                                   % (K, R, L),
                       "reads_per_gpu": R, "read_len": L, "K": K, "genome": cfg["genome"], "seed": seed,
                       "strategy": "both-strands" if a.both else "canonical+mirror",
                       **({"tune": os.environ["ZOT_TUNE"]} if os.environ.get("ZOT_TUNE") else {}),
                       "parallelism": "1 gpu" if world == 1 else "reads sharded over %d gpus + %s-owner all-to-all (%s)"
                                      % (world, par.owner, par.comm.name)},
            "roofline": {"bound": "hbm", "kernel": cands[dom][0],
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "peak_measured": peak_measured,
                         "peak_measured_note": "device-to-device copy of 4 GiB (read + written bytes per second), best of 5, this run",
                         "frac_of_measured": ach / peak_measured if peak_measured else None,
                         "traffic": traffic["bytes"] if traffic else None,
                         "traffic_note": ("HBM bytes per launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) of this command and these kernel sources, "
                                          + traffic["profile"]) if traffic else
                                         "null: no committed PMC profile matches the current kernel sources and workload (tools/collect_traffic.py)",
                         "launches": pk["launches"],
                         "avg_launch_ms": pk["ms"] / pk["launches"] if pk["launches"] else None,
                         "others": others},
            "pipeline": {"windows_per_s": value * 1e9 / 2, "instances_per_step": st.n_instances, "unique": st.n_unique,
                         "canonical_unique": st.n_canonical, "hist_bins": len(h),
                         # what the step really moves (PMC: 2 * FETCH_SIZE + WRITE_SIZE summed over every kernel of a step) over
                         # the step time: the whole step as a fraction of the HBM peak
                         "traffic_bytes_per_step": step_traffic["bytes"] if step_traffic else None,
                         "traffic_frac_of_peak": step_traffic["bytes"] / (dt / a.steps) / 1e9 / HBM_PEAK_GBS if step_traffic and world == 1 else None,
                         "traffic_note": step_traffic["profile"] if step_traffic else "null: no committed PMC profile of these kernel sources",
                         # SURVEY 8(d)'s UNFUSED model (both strands through every pass): the build no longer does that work, so this
                         # ratio is a speed-up over the naive plan at peak bandwidth, NOT a roofline fraction
                         "model_bytes": mb,
                         "speedup_over_unfused_model_at_peak": mb / (dt / a.steps) / 1e9 / HBM_PEAK_GBS if world == 1 else None,
                         "kernels": kernel_table(prof, a.steps)},
            "verified_checksums": verify, "verified_ascending": ascending,
            "extra": extras,
        }
        if owned is not None:
            out["owned_per_rank"] = owned
            out["balance_max_over_mean"] = max(owned) / (sum(owned) / float(len(owned)))
        if comm_note:
            out["comm_note"] = comm_note
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, seed)
        elif not a.no_cpu_baseline:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        if par is not None and hasattr(par.comm, "close"):
            par.comm.close()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
