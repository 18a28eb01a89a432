"""Run by tests/test_gpu_multigpu.py in its own process: the two RCCL transports with ONE rank on the GPU box (the box has one
GPU; the 8-GPU run is the driver's).  What one rank can prove on hardware:
  * zk_comm_* loads RCCL and creates a communicator.  With one rank zk_all_to_all_v is a device copy of the kept piece and
    zk_allreduce_u64 returns its input -- no RCCL call is made (`native_verified`, `native_allreduce`: the plumbing, no more);
  * the RCCL calls themselves, with ZK_TUNE_COMM_SELF_LOOP: the kept piece goes through grouped ncclSend / ncclRecv to the rank
    itself, in rounds of a forced small size (many rounds, a ragged last one, offsets that are not zero, empty pieces, 4- and
    8-byte elements), compared byte for byte with a plain device copy; zk_allreduce_u64 through ncclAllReduce (sum and max);
    and the product path -- Exchange.exchange_and_merge over that transport -- checked by the order-free checksums
    (`selfloop_*`).  What it cannot prove: the (me +- d) % W pairing and the offsets of OTHER ranks' pieces (world > 1);
  * torch.distributed "nccl": Exchange.exchange_and_merge with the round size forced small (many staged rounds), checked by
    the order-free checksums (zk_checksum of the merged table == zk_stream_checksum of the reads), as bench.py does;
  * the round-1 finding "one all_to_all_single above 1 GiB per peer arrives corrupt": a bare contiguous int64 tensor of
    2^27 + 2^20 elements through all_to_all_single vs a plain device copy -- isolates torch/RCCL from this module's slicing;
    the same payload through the chunked path must be intact.
Prints one JSON line (kept under profiles/ by the round's GPU run)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zotmer_amd import native, parallel, synth    # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
out = {}
ctx = native.Context(0)
K, R = 25, 200_000
kw = dict(genome=2_000_000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.0005))
stream = ctx.synth_reads(synth.DEFAULT_SEED, 0, R, 150, **kw)
want = ctx.stream_checksum(stream, K)
cap = 2 * stream.n
kt = torch.empty(cap, dtype=torch.int64, device="cuda")
ct = torch.empty(cap, dtype=torch.int32, device="cuda")
k, c, st = ctx.kmerize(stream, K, out=(native.DeviceArray.borrow(ctx, kt.data_ptr(), np.uint64, cap, keep=kt),
                                       native.DeviceArray.borrow(ctx, ct.data_ptr(), np.uint32, cap, keep=ct)))

# --- torch transport, many small rounds
ex = parallel.Exchange(ctx, dist, K, comm=parallel.TorchComm(dist))
ex.CHUNK = 50_000
ex.balanced_cuts([(kt, k.n)])
mk, mc = ex.exchange_and_merge(kt, ct, k.n)
out["torch_chunked_verified"] = bool(ex.verify_global(mk, mc, want)) and mk.n == k.n
out["torch_rounds"] = -(-k.n // ex.CHUNK)

# --- native transport (zk_comm_*)
comm = parallel.NativeComm(ctx, dist)
assert ctx.comm_info() == (1, 0)
exn = parallel.Exchange(ctx, dist, K, comm=comm, owner="hash", seed=5)
mk2, mc2 = exn.exchange_and_merge(kt, ct, k.n)
out["native_verified"] = bool(exn.verify_global(mk2, mc2, want)) and mk2.n == k.n
out["native_allreduce"] = comm.all_reduce([5, (1 << 64) - 1]) == [5, (1 << 64) - 1]
comm.close()

# --- the RCCL calls of comm.hip with the one rank there is: ncclSend / ncclRecv to self in rounds, ncclAllReduce
comm = parallel.NativeComm(ctx, dist, self_loop=True)
cases, bad_cases = 0, []
for eb, tdt in ((8, torch.int64), (4, torch.int32)):
    for n in (0, 1, 12345, (1 << 20) + 7):
        for chunk in (40_000, 1_000_003, 0):          # bytes per round: many rounds + a ragged tail; not a multiple of the element; the default
            ctx.tune(comm_chunk=chunk)
            soff, roff = 3, 5
            src_t = (torch.arange(n + soff + 2, dtype=torch.int64, device="cuda") * 2654435761 + 12345).to(tdt)
            dst_t = torch.full((n + roff + 2,), -7, dtype=tdt, device="cuda")
            ctx.all_to_all_v(src_t.data_ptr(), [soff], [n], dst_t.data_ptr(), [roff], [n], eb)
            ctx.sync()
            ok = bool(torch.equal(dst_t[roff:roff + n], src_t[soff:soff + n])) and bool((dst_t[:roff] == -7).all()) and bool((dst_t[roff + n:] == -7).all())
            cases += 1
            if not ok:
                bad_cases.append((eb, n, chunk))
out["selfloop_send_recv_cases"] = cases
out["selfloop_send_recv_bad"] = bad_cases
out["selfloop_rounds_max"] = -(-(8 * ((1 << 20) + 7)) // 40_000)
out["selfloop_allreduce"] = (comm.all_reduce([5, (1 << 64) - 1, 0]) == [5, (1 << 64) - 1, 0]
                             and comm.all_reduce([7, 1 << 63], "max") == [7, 1 << 63]
                             and list(comm.all_reduce(np.arange(1000, dtype=np.uint64))) == list(range(1000)))
ctx.tune(comm_chunk=300_000)
for owner in ("range", "hash"):
    exs = parallel.Exchange(ctx, dist, K, comm=comm, owner=owner, seed=5)
    if owner == "range":
        exs.balanced_cuts([(kt, k.n)])
    mk3, mc3 = exs.exchange_and_merge(kt, ct, k.n)
    out["selfloop_exchange_%s_verified" % owner] = bool(exs.verify_global(mk3, mc3, want)) and mk3.n == k.n
ctx.tune(comm_chunk=0)
comm.close()

# --- > 1 GiB per peer: bare all_to_all_single vs the chunked path, against a device copy
n_big = (1 << 27) + (1 << 20)
src = torch.arange(n_big, dtype=torch.int64, device="cuda") * 2654435761
dst = torch.zeros_like(src)
dist.all_to_all_single(dst, src, [n_big], [n_big])
torch.cuda.synchronize()
bad = int((dst != src).sum().item())
out["bare_all_to_all_single_gt_1GiB"] = {"elements": n_big, "bytes": 8 * n_big, "mismatching_elements": bad,
                                         "first_bad_index": int(torch.nonzero(dst != src)[0].item()) if bad else None}
dst.zero_()
tc = parallel.TorchComm(dist)
tc.all_to_all_v(dst, src, [n_big], [n_big], [0, n_big], [0, n_big])
torch.cuda.synchronize()
out["chunked_gt_1GiB_intact"] = bool(torch.equal(dst, src))
out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
out["torch"] = torch.__version__
dist.destroy_process_group()
ctx.close()
print("RCCL-ONE-RANK " + json.dumps(out))
ok = (out["torch_chunked_verified"] and out["native_verified"] and out["native_allreduce"] and out["chunked_gt_1GiB_intact"]
      and not out["selfloop_send_recv_bad"] and out["selfloop_allreduce"] and out["selfloop_exchange_range_verified"]
      and out["selfloop_exchange_hash_verified"])
sys.exit(0 if ok else 1)
