"""
Usage:
    zot dist [-M measure]... <k> <input>...

Options:
    -M measure  use "measure" for the distance between k-mer frequency sets.
                Use "-M list" to get a list of available measures.
"""
# Drop-in for zotmer/commands/dist.py: per file Measure.prep (dist.py:43-49) = prefix projection +
# adjacent dedupe on the device (zk_project_dedupe); per pair dist.split (library/dist.py:241-265) =
# zk_split; the closed-form measures stay on the host (library/measures.py).  Each file is decoded
# and uploaded once (the reference re-reads the right-hand file for every pair, dist.py:152-159).
import fnmatch
import sys

from zotmer_amd.library import engine, vectors
from zotmer_amd.library.container import KmerSet
from zotmer_amd.library.measures import MEASURES
from zotmer_amd.library.usage import Spec

_SPEC = Spec(options={"-M": "list"}, positionals=["<k>"], rest="<input>", rest_min=0)


class MismatchedK(Exception):       # zotmer/library/exceptions.py:31-38
    def __init__(self, k1, k2):
        self.k1, self.k2 = k1, k2

    def __str__(self):
        return "incompatible values of K: %d & %d" % (self.k1, self.k2)


def main(argv):
    opts = _SPEC.parse(argv[1:], __doc__)
    names = sorted(MEASURES)
    if "list" in opts["-M"]:                        # dist.py:97-104
        print("\n".join(m + "\t" + MEASURES[m][0] for m in names))
        return
    seen, bad = set(), False
    for pat in opts["-M"]:                          # dist.py:109-118
        hit = [m for m in names if fnmatch.fnmatch(m, pat)]
        if not hit:
            sys.stderr.write("warning: measure '%s' not found. Use -M list to see all measures.\n" % pat)
            bad = True
        seen.update(hit)
    ms = sorted(seen)
    if not ms or bad:                               # dist.py:122-123
        return
    vec = [m for m in ms if MEASURES[m][1]]
    if vec:
        raise SystemExit("zot dist: %s are spectrum (4**K counter) measures, which the reference cannot compute "
                         "either (see SURVEY.md appendix C.7); use the *.qual measures" % ", ".join(vec))

    K = int(opts["<k>"])
    files = opts["<input>"]
    dist, world, rank = engine.distributed()        # one process per GPU under torch.distributed.run
    ctx = engine.context()
    if dist is not None:
        return _main_distributed(ctx, dist, K, files, ms)

    def prep(path):                                 # dist.py:29-49
        with KmerSet(path, "r") as z:
            fK = z.meta["K"]
            if fK < K:
                raise MismatchedK(K, fK)
            k = vectors.device_read_kmers(ctx, z)
        return engine.measure_prep(ctx, k, 2 * (fK - K))

    print("\t".join(["lhs.name", "rhs.name"] + ms))
    sets = {}
    for i in range(len(files)):
        for j in range(i + 1, len(files)):
            for f in (files[i], files[j]):
                if f not in sets:
                    sets[f] = prep(f)
            abc = ctx.split(sets[files[i]], sets[files[j]])
            vals = [MEASURES[m][2](*abc) for m in ms]
            print("\t".join([files[i], files[j]] + ["%g" % v for v in vals]))


def _main_distributed(ctx, dist, K, files, ms):
    """Every rank decodes its own contiguous piece of each file (vectors.device_read_kmers_shard), the pieces of a pair
    are cut by one owner function and exchanged once, every rank splits what it owns, and (a, b, c) is all-reduced
    (zotmer_amd/parallel.py: Exchange.dist_pair).  Rank 0 prints."""
    import os
    import torch
    from zotmer_amd import parallel
    comm = parallel.make_comm(ctx, dist)
    ex = parallel.Exchange(ctx, dist, K, owner=os.environ.get("ZOT_OWNER", "range"), comm=comm)
    rank, world = comm.rank, comm.world

    def prep(path):
        with KmerSet(path, "r") as z:
            fK = z.meta["K"]
            if fK < K:
                raise MismatchedK(K, fK)
            k = vectors.device_read_kmers_shard(ctx, z, rank, world, comm)
        t = torch.empty(max(k.n, 1), dtype=torch.int64, device="cuda")
        if k.n:
            ctx._check(ctx.lib.zk_copy(ctx.h, t.data_ptr(), k.ptr, k.nbytes))
            ctx.sync()
        return t, k.n, 2 * (fK - K)

    if rank == 0:
        print("\t".join(["lhs.name", "rhs.name"] + ms))
    sets = {}
    for i in range(len(files)):
        for j in range(i + 1, len(files)):
            for f in (files[i], files[j]):
                if f not in sets:
                    sets[f] = prep(f)
            (xt, nx, sx), (yt, ny, sy) = sets[files[i]], sets[files[j]]
            if sx != sy:                            # different file K: project each side on its own first
                xt, nx = ex.ops.dedupe(xt, nx, sx)
                yt, ny = ex.ops.dedupe(yt, ny, sy)
                sx = 0
            abc, _ = ex.dist_pair(xt, nx, yt, ny, shift=sx)
            if rank == 0:
                vals = [MEASURES[m][2](*abc) for m in ms]
                print("\t".join([files[i], files[j]] + ["%g" % v for v in vals]))


if __name__ == "__main__":
    main(["dist"] + sys.argv[1:])
