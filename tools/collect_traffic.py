#!/usr/bin/env python3
"""Turn two rocprofv3 PMC runs of bench.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, each
also with --kernel-trace, CSV output) into the per-launch HBM traffic of EVERY kernel of the step, applying
the gfx950 corrections of MI355X_MICROARCH.md (HBM section): the counters are in KiB; FETCH_SIZE
reports half of the bytes of a wide coalesced read stream and is doubled; WRITE_SIZE is exact.

usage: collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <keys per step> <steps profiled>"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


f, w = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
n_keys = int(sys.argv[4]) if len(sys.argv) > 4 else None
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
out = {}
total = 0.0
for name in sorted(set(f) | set(w)):
    if any(x in name for x in ("checksum", "synth", "descent")):          # the bench's generator and checkers, not the step
        continue
    fv, wv = f.get(name, [0.0]), w.get(name, [0.0])
    fk, wk = sum(fv) / len(fv), sum(wv) / len(wv)
    launches = max(len(fv), len(wv))
    t = (2 * fk + wk) * 1024
    total += t * launches
    if t * launches < 1e6:          # setup kernels: nothing to report
        continue
    out[name] = dict(launches=launches, fetch_size_kib=fk, write_size_kib=wk, traffic_bytes_per_launch=t, n_keys=n_keys)
out["_traffic_bytes_per_step"] = total / steps
out["_steps_profiled"] = steps
import bench     # the hash of the kernel sources the profile was taken from: bench.py quotes a row only while it matches
out["_kernel_source_sha256"] = bench.kernel_source_hash()
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"]) if isinstance(kv[1], dict) else 0):
    if isinstance(v, dict):
        print("%-110s %3d launches  %8.2f GB/launch" % (k[:110], v["launches"], v["traffic_bytes_per_launch"] / 1e9))
print("all kernels: %.1f GB per step" % (total / steps / 1e9))
