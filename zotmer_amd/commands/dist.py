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
    ctx = engine.context()

    def prep(path):                                 # dist.py:29-49
        with KmerSet(path, "r") as z:
            fK = z.meta["K"]
            if fK < K:
                raise MismatchedK(K, fK)
            k = vectors.device_read_kmers(ctx, z)
        return ctx.project_dedupe(k, 2 * (fK - K))

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


if __name__ == "__main__":
    main(["dist"] + sys.argv[1:])
