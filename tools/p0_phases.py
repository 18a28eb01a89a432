#!/usr/bin/env python3
"""Diagnostic: where wave 0 of a stream_pass0_kernel workgroup spends a tile (in-kernel s_memtime, 100 MHz ticks), and the
HIP-event times of the histogram and the pass; the same for dedupe_kernel per block.  Needs the diagnostic build:
make -C zotmer_amd/csrc clean && make -C zotmer_amd/csrc CXXFLAGS_EXTRA=-DZK_PHASES (rebuild without it afterwards).  usage: p0_phases.py [reads] [stream_pass variant] [K]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zotmer_amd import native, synth

reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cfg = synth.CONFIGS["config2"]
ctx = native.Context(0)
ctx.tune(stream_pass=variant)
if os.environ.get("ZOT_TUNE"):          # e.g. ZOT_TUNE=dedupe_variant=1
    ctx.tune(**{k: int(v) for k, v in (kv.split("=") for kv in os.environ["ZOT_TUNE"].split(","))})
d = ctx.synth_reads(synth.DEFAULT_SEED, 0, reads, cfg["L"], genome=cfg["genome"], sub_thr=synth.frac32(cfg["sub"]), n_thr=synth.frac32(cfg["n"]))
cap = int(2 * (min(cfg["genome"], reads * cfg["L"]) + reads * cfg["L"] * cfg["sub"] * 22) * 1.25) + (1 << 20)
outs = (ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32))
ctx.kmerize(d, K, out=outs)
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 1: the pass without its stores, 2: the stores of a range's first tile only, repeated
ranges = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ctx.tune(stream_pass=variant | (mode << 8), stream_ranges=ranges)
dbg = ctx.upload(np.zeros(4096 * 16, np.uint64))
ctx._check(ctx.lib.zk_debug_buffer(ctx.h, dbg.ptr))
ctx.profile(True)
ctx.kmerize(d, K, out=outs)
ctx.sync()
prof = ctx.profile_read()
ctx.profile(False)
ctx._check(ctx.lib.zk_debug_buffer(ctx.h, None))
raw = dbg.to_host().reshape(4096, 16).astype(np.float64)
raw = raw[raw[:, 8] > 0]
names = ["0 keys + rank", "1 barrier", "2 scan (2 barriers)", "3 park", "4 barrier", "5 next image (+ wait for bytes)", "6 stores issued", "7 barrier"]
tiles = raw[:, 8].sum()
per = raw[:, :8].sum(axis=0) / tiles * 10.0          # ns per tile
out = {"reads": reads, "variant": variant, "mode": mode, "K": K, "ranges": int(len(raw)), "tiles_per_range": float(raw[:, 8].mean()),
       "ns_per_tile": {n: round(float(v), 1) for n, v in zip(names, per)}, "ns_per_tile_total": round(float(per.sum()), 1),
       "hist_stream_ms": prof.get("hist_stream", {}).get("ms"), "pass_stream_ms": prof.get("pass_stream", {}).get("ms")}
dd = dbg.to_host().reshape(4096, 16).astype(np.float64)[512:512 + 256]          # dedupe_kernel's rows (zk_debug_buffer + 8192 words)
dd = dd[dd[:, 8] > 0]
if len(dd):
    nb = dd[:, 8].sum()
    names = ["0 clear table", "1 insert keys (+ loads)", "2 drain + barrier", "3 read entries, count byte groups", "4 scan + group", "5 rank + write", "6 barrier"]
    out["dedupe_cycles_per_block"] = {n: round(float(v), 1) for n, v in zip(names, dd[:, :7].sum(axis=0) / nb)}
    out["dedupe_blocks"] = int(nb)
    out["rle_ms"] = prof.get("rle", {}).get("ms")
print(json.dumps(out))
