#!/usr/bin/env python3
"""
bench.py -- `zot kmerize` K=25 on synthetic 150 bp reads, the metric of BASELINE.json.

  python bench.py --gpus N --steps K --warmup W
  (N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step is one whole kmerize of the batch: base stream resident in HBM -> sorted distinct k-mers
of both strands + counts in HBM (hist and acgt included), i.e. zk_kmerize + zk_hist through the
C-ABI.  At N = 1 the workload is BASELINE config 2 (50 M x 150 bp genome-sampled reads, K = 25).
At N > 1 every rank kmerizes its own 50 M reads (weak scaling) and the per-rank tables are then
exchanged by k-mer value range with one RCCL all-to-all and union-summed, so that each rank ends
up owning one contiguous range of the global table.

One JSON line on stdout (rank 0).  `value` counts emitted k-mer instances (both strands, the unit
the reference counts at commands/kmerize.py:523-525) per second of wall time over the timed steps.
`roofline` is the dominant kernel (one radix-sort pass over the key array): algorithmic 16 B/key
over its mean launch time, measured with HIP events on the library's own stream.  `cpu_baseline` is
the CPU oracle (a single-core C restatement of the reference algorithm) on a bounded sample of the
same reads.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy reaches


def model_bytes(n_stream_bytes, instances, unique, K):
    """unfused per-stage byte model the fraction is always quoted against."""
    P = -(-2 * K // 8)
    return n_stream_bytes + 8 * instances + (1 + 2 * P) * 8 * instances + 8 * instances + 12 * unique


def measured_traffic(kernel_substr, n_keys_now):
    """HBM bytes per launch of the dominant kernel from the committed PMC runs of THIS command
    (profiles/r01_config2_pipeline/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950; tools/collect_traffic.py).  PMC
    counters cannot be read from inside the timed run, so the figure is taken from that profile and only
    reported when the workload (number of keys per launch) is the profiled one."""
    path = os.path.join(ROOT, "profiles", "r01_config2_pipeline", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    for name, v in d.items():
        if kernel_substr in name and abs(v.get("n_keys", 0) - n_keys_now) <= 0.001 * max(n_keys_now, 1):
            return v["traffic_bytes_per_launch"]
    return None


def cpu_baseline(cfg, seed, budget_s=12.0):
    """Single-core CPU oracle on a prefix of the same reads, sized to about budget_s seconds."""
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
    return {"value": inst / t / 1e9, "unit": "Gk-mers/s", "cores": 1, "kind": "port",
            "sample": "first %d of the %d reads (same generator, same K), C oracle oracle/zk_oracle.c: per-read window loop, "
                      "MSD radix + qsort, RLE merge; %.1f s" % (n, cfg["reads"], t)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: config 2 = 50 M)")
    ap.add_argument("--both", action="store_true", help="sort both strands literally instead of canonical + mirror")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true", help="check the order-free checksums of the result against the stream")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("ZOT_FORCE_EXCHANGE") == "1":
        # torch bundles its own HIP runtime: let it load first so that libzotk.so binds to the same
        # copy (same SONAME) instead of dragging a second runtime into the process
        import torch  # noqa: F401
    # a rebuild (make) or RCCL's version banner may write to stdout; the contract is ONE JSON line there,
    # so everything but the final print goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    from zotmer_amd import native, synth
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")

    dist = None
    force_exchange = os.environ.get("ZOT_FORCE_EXCHANGE") == "1"      # rehearse the N > 1 path with one rank
    if world > 1 or force_exchange:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    cfg = dict(synth.CONFIGS["config2"])
    if a.reads:
        cfg["reads"] = a.reads
    R, L, K = cfg["reads"], cfg["L"], cfg["K"]
    seed = synth.DEFAULT_SEED
    flags = native.KMERIZE_BOTH if a.both else native.KMERIZE_CANONICAL

    ctx = native.Context(local)
    n_bytes = R * (L + 1)
    stream = ctx.synth_reads(seed, rank * R, R, L, genome=cfg["genome"], sub_thr=synth.frac32(cfg["sub"]),
                             n_thr=synth.frac32(cfg["n"]))
    ctx.sync()
    # distinct k-mers: 2 strands x (genome + ~21 novel windows per substitution); leave headroom
    est_unique = int(2 * (min(cfg["genome"], R * L) + R * L * cfg["sub"] * 22) * 1.25) + (1 << 20)
    cap = min(est_unique, 2 * n_bytes)
    par = None
    if dist is not None:
        # the exchange goes through torch.distributed, so the table lives in torch tensors that the
        # library writes through their data_ptr() (uint64 carried as int64, uint32 as int32)
        import torch
        from zotmer_amd import parallel
        kt = torch.empty(cap, dtype=torch.int64, device="cuda")
        ct = torch.empty(cap, dtype=torch.int32, device="cuda")
        out_k = native.DeviceArray.borrow(ctx, kt.data_ptr(), np.uint64, cap, keep=kt)
        out_c = native.DeviceArray.borrow(ctx, ct.data_ptr(), np.uint32, cap, keep=ct)
        par = parallel.RangeExchange(ctx, dist, K)
    else:
        out_k, out_c = ctx.empty(cap, np.uint64), ctx.empty(cap, np.uint32)

    def step():
        k, c, st = ctx.kmerize(stream, K, flags, out=(out_k, out_c))
        if par is not None:
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

    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([st.n_instances], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        total_instances = int(tot.item())
    else:
        total_instances = st.n_instances

    verify = None
    if a.verify:
        want = ctx.stream_checksum(stream, K)
        if par is None:
            got = ctx.checksum(k, c)
            verify = bool(got == want)
        else:
            verify = par.verify_global(k, c, want)

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = total_instances * a.steps / dt / 1e9
        pk = prof.get("pass_keys", dict(launches=0, ms=0.0, bytes=0))
        ach = (pk["bytes"] / 1e9) / (pk["ms"] / 1e3) if pk["ms"] else 0.0
        mb = model_bytes(n_bytes, st.n_instances, st.n_unique, K)
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
                       "parallelism": "1 gpu" if world == 1 else "reads sharded over %d gpus + value-range all-to-all" % world},
            "roofline": {"bound": "hbm", "kernel": "pass_pipe_kernel<array,keys> (one LSD radix pass of 64-bit keys, 16 B/key, persistent two-stage pipeline)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": measured_traffic("pass_pipe_kernel<zk::Cfg<512, 16, 9, 1, 4, 32, true>, 0>", st.n_windows),
                         "traffic_note": "HBM bytes per launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, KiB) of this "
                                         "command, profiles/r01_config2_pipeline/pmc_traffic.json; null if the workload differs",
                         "launches": pk["launches"],
                         "avg_launch_ms": pk["ms"] / pk["launches"] if pk["launches"] else None},
            "pipeline": {"windows_per_s": value * 1e9 / 2, "instances_per_step": st.n_instances, "unique": st.n_unique,
                         "canonical_unique": st.n_canonical, "model_bytes": mb,
                         "model_frac_of_peak": mb / (dt / a.steps) / 1e9 / HBM_PEAK_GBS if world == 1 else None,
                         "hist_bins": len(h),
                         "kernels": {n: dict(launches=v["launches"], ms_per_step=v["ms"] / a.steps,
                                             GBps=(v["bytes"] / 1e9) / (v["ms"] / 1e3) if v["ms"] else None)
                                     for n, v in prof.items()}},
        }
        if verify is not None:
            out["verified_checksums"] = verify
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, seed)
        elif not a.no_cpu_baseline:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
