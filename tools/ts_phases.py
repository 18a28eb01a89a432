#!/usr/bin/env python3
"""Diagnostic: where thread 0 of a tile_sort_count_kernel workgroup spends a tile (in-kernel s_memtime, 100 MHz ticks) on reads that
do not repeat their k-mers.  Needs the diagnostic build (tools/build_phases.sh, ZOTK_LIB=build/libzotk_phases.so).
usage: ts_phases.py [reads] [K]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zotmer_amd import native, synth

reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ctx = native.Context(0)
if os.environ.get("ZOT_TUNE"):
    ctx.tune(**{k: int(v) for k, v in (kv.split("=") for kv in os.environ["ZOT_TUNE"].split(","))})
d = ctx.synth_reads(synth.DEFAULT_SEED, 0, reads, 150, genome=0, n_thr=synth.frac32(0.0005))
cap = reads * (150 - K + 1) + (1 << 20)
outs = (ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32))
flags = native.KMERIZE_CANONICAL_ONLY
ctx.kmerize(d, K, flags, out=outs)
dbg = ctx.upload(np.zeros(4096 * 16, np.uint64))
ctx._check(ctx.lib.zk_debug_buffer(ctx.h, dbg.ptr))
ctx.profile(True)
k, c, st = ctx.kmerize(d, K, flags, out=outs)
ctx.sync()
prof = ctx.profile_read()
ctx.profile(False)
ctx._check(ctx.lib.zk_debug_buffer(ctx.h, None))
raw = dbg.to_host().reshape(4096, 16).astype(np.float64)[:256]
raw = raw[raw[:, 8] > 0]
names = ["0 ticket", "1 load + group", "2 rank", "3 back to LDS sorted", "4 heads", "5 look-back + barrier", "6 write"]
tiles = raw[:, 8].sum()
per = raw[:, :7].sum(axis=0) / tiles * 10.0
print(json.dumps({"reads": reads, "K": K, "n_unique": k.n, "workgroups": int(len(raw)), "tiles_per_workgroup": float(raw[:, 8].mean()),
                  "ns_per_tile": {n: round(float(v), 1) for n, v in zip(names, per)}, "ns_per_tile_total": round(float(per.sum()), 1),
                  "kernels_ms": {n: round(v["ms"], 2) for n, v in prof.items() if v["launches"]}}))
