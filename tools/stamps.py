#!/usr/bin/env python3
"""Diagnostic: per-phase time of pass_kernel from in-kernel s_memtime stamps.
Build first with:  make -C zotmer_amd/csrc clean && make -C zotmer_amd/csrc CXXFLAGS_EXTRA=-DZK_STAMPS
Shares, not absolute times, are what to read (the stamped build is slower)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zotmer_amd import native

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tile = {0: 8192, 3: 8192, 1: 4096, 2: 8192, 4: 8192, 5: 8192, 6: 8192, 7: 8192}[variant]
ctx = native.Context(0)
ctx.tune(sort_variant=variant)
rng = np.random.default_rng(1)
keys = ctx.upload(rng.integers(0, 1 << 50, size=n, dtype=np.uint64))
tiles = (n + tile - 1) // tile
dbg = ctx.empty(tiles * 17, np.uint64)
ctx.sort_keys(keys, 50)           # warm
keys = ctx.upload(rng.integers(0, 1 << 50, size=n, dtype=np.uint64))
ctx._check(ctx.lib.zk_debug_buffer(ctx.h, dbg.ptr))
ctx.sort_keys(keys, 50)           # the stamps of the LAST pass remain
ctx._check(ctx.lib.zk_debug_buffer(ctx.h, None))
raw = dbg.to_host()
s = raw[:tiles * 8].reshape(tiles, 8).astype(np.int64)
ss = raw[tiles * 8:tiles * 9]
per_wave = raw[tiles * 9:tiles * 17].reshape(tiles, 8).astype(np.int64)
steps = (ss >> np.uint64(32)).astype(np.int64)[1:]
spins = (ss & np.uint64(0xFFFFFFFF)).astype(np.int64)[1:]
d = np.diff(s, axis=1)
pipe = variant == 3
if pipe:
    ok = (s[:, 7] > 0) & (s[:, 0] > 0)
    s = s[ok]
    seq = [("B: load wait", 0, 1), ("B: rank (wave 0)", 1, 2), ("B: barrier + scan + publish", 2, 4),
           ("(A's offsets, barrier, A's store, barrier) + park B", 4, 3), ("parked until offsets known", 3, 5),
           ("A: barrier after offsets", 5, 6), ("A: store issue", 6, 7)]
    tot = s[:, 7] - s[:, 0]
    pw = per_wave[ok]
    good = (pw > 0).all(axis=1)
    rel = pw[good] - pw[good].min(axis=1, keepdims=True)
    print(json.dumps({"per_wave_offsets_known_minus_first_median": [float(np.median(rel[:, w])) for w in range(8)],
                      "per_wave_is_last_fraction": [float((rel.argmax(axis=1) == w).mean()) for w in range(8)]}))
    print(json.dumps({"hops_mean": float(steps.mean()), "polls_mean": float(spins.mean()), "polls_p50_p90_p99": [float(np.percentile(spins, q)) for q in (50, 90, 99)],
                      "tiles": int(len(s)), "median_ticks_ticket_to_stored": float(np.median(tot)),
                      "phases_median_ticks": {nm: float(np.median(s[:, b] - s[:, a])) for nm, a, b in seq}}, indent=1))
    sys.exit(0)
names = ["load wait", "rank (wave 0)", "barrier after rank", "digit scan", "look-back (thread 0)", "barrier after look-back", "regroup + store"]
tot = (s[:, 7] - s[:, 0])
print(json.dumps({"steps_mean": float(steps.mean()), "steps_p10_p50_p90_p99": [float(np.percentile(steps, q)) for q in (10, 50, 90, 99)],
                  "ticks_per_step_median": float(np.median(d[1:, 4] / np.maximum(steps, 1))), "spins_mean": float(spins.mean()), "spins_p50_p90_p99": [float(np.percentile(spins, q)) for q in (50, 90, 99)],
                  "frac_tiles_with_spin": float((spins > 0).mean()),
                  "tiles": int(tiles), "median_total_ticks": float(np.median(tot)),
                  "phases_median_ticks": {nm: float(np.median(d[:, i])) for i, nm in enumerate(names)},
                  "phases_mean_share": {nm: float(d[:, i].sum() / tot.sum()) for i, nm in enumerate(names)}}, indent=1))
