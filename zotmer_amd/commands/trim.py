"""
Usage:
    zot trim [-c CUTOFF] [-C CUTOFF] <output> <input>

Options:
    -c CUTOFF   discard k-mers with frequency less than CUTOFF. A
                cutoff of 0 (the default) indicates that cutoff
                inference should be used. [default: 0]
    -C CUTOFF   discard k-mers with frequency greater than CUTOFF.
                A cutoff of 0 (the default) indicates that the
                cutoff value should be effectively infinite.
                [default: 0]
"""
# Drop-in for zotmer/commands/trim.py: the filter (trim.py:54-62) is zk_trim on the device.  The
# metadata is copied from the input, `hist` included and NOT recomputed, exactly as trim.py:88-95.
# Cutoff inference (trim.py:22-52) works here; in the reference it raises TypeError because the
# histogram keys come back from JSON as strings.
import math
import sys

from zotmer_amd.library import engine, vectors
from zotmer_amd.library.container import KmerSet
from zotmer_amd.library.usage import Spec

_SPEC = Spec(options={"-c": True, "-C": True}, positionals=["<output>", "<input>"])


def infer(hist):
    """First local minimum of the Gaussian-smoothed (sigma 1.5) count histogram (trim.py:22-52)."""
    h = {int(k): v for k, v in hist.items()}
    if not h:
        return 0
    sm = {}
    for x0 in h:
        num = den = 0.0
        for xi, yi in h.items():
            w = math.exp(-((x0 - xi) * (x0 - xi)) / (2.0 * 1.5 * 1.5))
            num += w * yi
            den += w
        sm[x0] = num / den
    items = sorted(sm.items())
    best = items[0]
    for it in items[1:]:
        if it[1] < best[1]:
            best = it
        else:
            break
    return best[0]


def main(argv):
    opts = _SPEC.parse(argv[1:], __doc__)
    c = int(opts["-c"]) if opts["-c"] is not None else 0
    C = int(opts["-C"]) if opts["-C"] is not None else 0
    ctx = engine.context()
    with KmerSet(opts["<input>"], "r") as z:
        meta = dict(z.meta)
        if c == 0:
            c = infer(meta.get("hist", {}))
            sys.stderr.write("inferred cutoff: %d\n" % c)        # trim.py:85
        k, cn = vectors.device_read_kmers_and_counts(ctx, z)
    tk, tc = ctx.trim(k, cn, c, C if C > 0 else 0)
    with KmerSet(opts["<output>"], "w") as w:
        vectors.device_write_kmers_and_counts(ctx, w, tk, tc)
        w.meta = meta
        w.meta["kmers"] = "kmers"
        w.meta["counts"] = "counts"


if __name__ == "__main__":
    main(["trim"] + sys.argv[1:])
