#!/usr/bin/env python3
"""Turn two rocprofv3 PMC runs of bench.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, each
also with --kernel-trace, CSV output) into the per-launch HBM traffic of the dominant kernel, applying
the gfx950 corrections of MI355X_MICROARCH.md (HBM section): the counters are in KiB; FETCH_SIZE
reports half of the bytes of a wide coalesced read stream and is doubled; WRITE_SIZE is exact.

usage: collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [keys per array pass]"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg

f, w = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
out = {}
for name in f:
    if name in w and ("pass_kernel" in name or "pass_pipe_kernel" in name or "rle_kernel" in name or "union_sum" in name or "hist_kernel" in name):
        fk, wk = sum(f[name]) / len(f[name]), sum(w[name]) / len(w[name])
        out[name] = dict(launches=len(f[name]), fetch_size_kib=fk, write_size_kib=wk,
                         traffic_bytes_per_launch=(2 * fk + wk) * 1024,
                         n_keys=(int(sys.argv[4]) if len(sys.argv) > 4 and "pass_pipe_kernel" in name and (name.rstrip().endswith(", 0, 0>(zk::SortArgs, unsigned int)") or name.rstrip().endswith(", 1, 0>(zk::SortArgs, unsigned int)")) else None))
import bench     # the hash of the kernel sources the profile was taken from: bench.py quotes a row only while it matches
out["_kernel_source_sha256"] = bench.kernel_source_hash()
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print("%-100s %3d launches  %.2f GB/launch" % (k[:100], v["launches"], v["traffic_bytes_per_launch"] / 1e9))
