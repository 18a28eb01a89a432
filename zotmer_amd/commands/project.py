"""
Usage:
    zot project <ref> <output> <input>

Project one or more inputs on to a reference set. For each k-mer in <ref>,
a whitespace separated 0 or a 1 is printed indicating whether that k-mer
was present in the input, with a separate line for each input k-mer set.
"""
# Drop-in for zotmer/commands/project.py.  What the command does (whatever its usage text says) is
# write the entries of <input> whose k-mer is in <ref> (project1/project2, project.py:16-40): one
# merge-path pass on the device (zk_project).  `hist` is copied from the input, not recomputed
# (project.py:66); an input without counts gives an output without counts (project.py:62-65).
import sys

import numpy as np

from zotmer_amd.library import engine, vectors
from zotmer_amd.library.container import KmerSet
from zotmer_amd.library.usage import Spec

_SPEC = Spec(positionals=["<ref>", "<output>", "<input>"])


def main(argv):
    opts = _SPEC.parse(argv[1:], __doc__)
    ctx = engine.context()
    with KmerSet(opts["<ref>"], "r") as z:
        K = z.meta["K"]
        ref = vectors.device_read_kmers(ctx, z)
    with KmerSet(opts["<input>"], "r") as z0:
        K0 = z0.meta["K"]
        if K0 != K:
            sys.stderr.write("mismatched K (%d)\n" % K0)             # project.py:50-52
            raise SystemExit(1)
        with KmerSet(opts["<output>"], "w") as z:
            z.meta["K"] = K
            if "counts" in z0.meta:
                k, c = vectors.device_read_kmers_and_counts(ctx, z0)
                pk, pc = ctx.project(ref, k, c)
                vectors.device_write_kmers_and_counts(ctx, z, pk, pc)
                z.meta["kmers"] = "kmers"
                z.meta["counts"] = "counts"
            else:
                k = vectors.device_read_kmers(ctx, z0)
                pk, _ = ctx.project(ref, k, ctx.upload(np.zeros(k.n, np.uint64)))
                z.add("kmers", vectors.device_encode_kmers(ctx, pk))
                z.meta["kmers"] = "kmers"
            z.meta["hist"] = z0.meta["hist"]


if __name__ == "__main__":
    main(["project"] + sys.argv[1:])
