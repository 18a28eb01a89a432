"""
Usage:
    zot merge <output> <input>...
"""
# Drop-in for zotmer/commands/merge.py.  The pairwise streaming merge into a temporary container
# followed by mergeNinto (merge.py:201-251) becomes one device k-way union-sum (zk_merge_n).
# Unlike the reference, which writes only hist/acgt for one or two inputs (merge.py:173-199, and
# fails outright for one), the full metadata is written in every case; the arrays are identical.
# Under `python -m torch.distributed.run --nproc-per-node N ... zot merge ...` the sets are sharded over N GPUs
# (rank r loads sets r, r+N, ...), exchanged once by k-mer owner over RCCL, and rank 0 writes the same file.
import sys

from zotmer_amd.library import engine, vectors
from zotmer_amd.library.container import KmerSet
from zotmer_amd.library.usage import Spec

_SPEC = Spec(positionals=["<output>"], rest="<input>")


def main(argv):
    opts = _SPEC.parse(argv[1:], __doc__)
    out, inputs = opts["<output>"], opts["<input>"]
    dist, world, rank = engine.distributed()        # one process per GPU under torch.distributed.run
    ctx = engine.context()
    K = None
    sets = []
    for path in inputs[rank::world]:                # rank r loads sets r, r + world, ...
        with KmerSet(path, "r") as z:
            k0 = z.meta["K"]
            if K is None:
                K = k0
            elif k0 != K:                           # merge.py:186-190
                sys.stderr.write("mismatched K\n")
                sys.exit(1)
            sets.append(vectors.device_read_kmers_and_counts(ctx, z))     # decoded on the device
    if dist is not None:
        mk, mc, acgt, hist = _merge_distributed(ctx, dist, K, sets)
        if rank != 0:
            return
    else:
        mk, mc, acgt = ctx.merge_n(sets)
        hist = ctx.hist(mc)
    with KmerSet(out, "w") as z:
        vectors.device_write_kmers_and_counts(ctx, z, mk, mc)
        total = float(sum(acgt))
        z.meta["K"] = K
        z.meta["kmers"] = "kmers"
        z.meta["counts"] = "counts"
        z.meta["hist"] = hist
        z.meta["acgt"] = [c / total for c in acgt]  # count-weighted (merge.py:159,245-246)


def _merge_distributed(ctx, dist, K, sets):
    """The ranks' local merges meet in one exchange (zotmer_amd/parallel.py: Exchange.merge_sets); rank 0 gathers the
    owned pieces and writes the file.  -> (k-mers, counts, acgt, hist) on rank 0."""
    import os
    from zotmer_amd import parallel
    comm = parallel.make_comm(ctx, dist)
    ks = comm.all_gather_object(K)
    ks = [k for k in ks if k is not None]
    if len(set(ks)) > 1:
        if dist.get_rank() == 0:
            sys.stderr.write("mismatched K\n")
        sys.exit(1)
    K = ks[0]
    ex = parallel.Exchange(ctx, dist, K, owner=os.environ.get("ZOT_OWNER", "range"), comm=comm)
    if sets:
        k, c, _ = ctx.merge_n(sets)
    else:
        k, c = ctx.empty(0, "u8").view(0), ctx.empty(0, "u8").view(0)
    kt, ct, n = ex.ops.to_tensors(k, c)
    res = ex.merge_sets(kt, ct, n)
    gk, gc = ex.gather_to_root(res["k"], res["c"])
    return gk, gc, res["acgt"], res["hist"]


if __name__ == "__main__":
    main(["merge"] + sys.argv[1:])
