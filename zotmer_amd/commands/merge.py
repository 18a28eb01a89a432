"""
Usage:
    zot merge <output> <input>...
"""
# Drop-in for zotmer/commands/merge.py.  The pairwise streaming merge into a temporary container
# followed by mergeNinto (merge.py:201-251) becomes one device k-way union-sum (zk_merge_n).
# Unlike the reference, which writes only hist/acgt for one or two inputs (merge.py:173-199, and
# fails outright for one), the full metadata is written in every case; the arrays are identical.
import sys

from zotmer_amd.library import engine, vectors
from zotmer_amd.library.container import KmerSet
from zotmer_amd.library.usage import Spec

_SPEC = Spec(positionals=["<output>"], rest="<input>")


def main(argv):
    opts = _SPEC.parse(argv[1:], __doc__)
    out, inputs = opts["<output>"], opts["<input>"]
    ctx = engine.context()
    K = None
    sets = []
    for path in inputs:
        with KmerSet(path, "r") as z:
            k0 = z.meta["K"]
            if K is None:
                K = k0
            elif k0 != K:                           # merge.py:186-190
                sys.stderr.write("mismatched K\n")
                sys.exit(1)
            sets.append(vectors.device_read_kmers_and_counts(ctx, z))     # decoded on the device
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


if __name__ == "__main__":
    main(["merge"] + sys.argv[1:])
