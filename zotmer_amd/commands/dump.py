"""
Usage:
    zot dump <input>
"""
# zotmer/commands/dump.py: `kmer<TAB>count` per line (or just the k-mer when the set has no counts).
import sys

import numpy as np

from zotmer_amd.library import vectors
from zotmer_amd.library.container import KmerSet
from zotmer_amd.library.usage import Spec

_SPEC = Spec(positionals=["<input>"])
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def render(K, kmers):
    """basics.render (zotmer/library/basics.py:61-67), vectorised: uint64[n] -> list of str"""
    k = np.asarray(kmers, dtype=np.uint64)
    shifts = (2 * np.arange(K - 1, -1, -1, dtype=np.uint64))[None, :]
    codes = ((k[:, None] >> shifts) & np.uint64(3)).astype(np.intp)
    return [row.tobytes().decode() for row in _ACGT[codes]]


def main(argv):
    path = _SPEC.parse(argv[1:], __doc__)["<input>"]
    with KmerSet(path, "r") as z:
        K = z.meta["K"]
        if "kmers" not in z.meta:
            sys.stderr.write('cannot dump "%s" as it contains no k-mers\n' % path)
            return
        if "counts" in z.meta:
            k, c = vectors.read_kmers_and_counts(z)
        else:
            k, c = vectors.read_kmers(z), None
    out = sys.stdout
    step = 1 << 16
    for i in range(0, len(k), step):
        names = render(K, k[i:i + step])
        if c is None:
            out.write("\n".join(names) + "\n")
        else:
            out.write("".join("%s\t%d\n" % (s, n) for s, n in zip(names, c[i:i + step])))


if __name__ == "__main__":
    main(["dump"] + sys.argv[1:])
