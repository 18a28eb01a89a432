"""
Usage:
    zot jaccard [-abp P] <input>...

Compute Jaccard indexes between k-mer sets. By default, indexes are
computed only between the first k-mer set and all the remaining
k-mer sets. If the -a option is given, all pairwise indexes are
computed.  If the -p P option is given, a Null hypothesis test is
performed for the hypothesis that the underlying Jaccard Index is
less than P. This is particularly useful if subsets of k-mers are
being used (NB, if the k-mer sets are large, the statistics can be
very expensive to compute).

Options:
    -a          print all pairwise distances
    -p P        Jaccard distance thresshhold for p-value computation
"""
# Drop-in for zotmer/commands/jaccard.py.  The two-cursor walk (jaccard.py:30-54) is zk_split on the
# device; a single FASTA input (jaccard.py:108-136) turns every record into its sorted set of
# both-strand 25-mers with zk_kmerize; the beta-function statistics stay on the host
# (library/jstats.py).  Every set is decoded and uploaded once.
import sys

from zotmer_amd import native
from zotmer_amd.library import engine, seqio, vectors
from zotmer_amd.library.container import KmerSet
from zotmer_amd.library.jstats import jaccard_fields
from zotmer_amd.library.usage import Spec

_SPEC = Spec(options={"-a": False, "-b": False, "-p": True}, positionals=[], rest="<input>")


def _pairs(names, sets, all_pairs, p):
    for i in range(len(names) if all_pairs else 1):
        for j in range(i + 1, len(names)):
            x, y = sets(i), sets(j)
            isec = engine.context().split(x, y)[0]
            print("%s\t%s\t%s" % (names[i], names[j], jaccard_fields(x.n, y.n, isec, p)))
            sys.stdout.flush()


def main(argv):
    opts = _SPEC.parse(argv[1:], __doc__)
    files = opts["<input>"]
    p = float(opts["-p"]) if opts["-p"] is not None else None
    ctx = engine.context()

    if len(files) == 1 and seqio.is_fasta(files[0]):
        K = 25
        names, recs = [], []
        for nm, seq in seqio.fasta_records(files[0]):
            d = ctx.upload_stream(seq + b"\n")
            k, c, _ = ctx.kmerize(d, K, native.KMERIZE_BOTH)
            names.append(nm.split()[0])
            recs.append(ctx.copy_of(k))
        print(len(recs))
        _pairs(names, lambda i: recs[i], opts["-a"], p)
        return

    cache, K0 = {}, []

    def load(i):
        if i not in cache:
            with KmerSet(files[i], "r") as z:
                K = z.meta["K"]
                cache[i] = vectors.device_read_kmers(ctx, z)
            K0.append(K)
            if K != K0[0]:
                sys.stderr.write("mismatched K: %s\n" % files[i])        # jaccard.py:151-153
                raise SystemExit(1)
        return cache[i]

    _pairs(files, load, opts["-a"], p)


if __name__ == "__main__":
    main(["jaccard"] + sys.argv[1:])
