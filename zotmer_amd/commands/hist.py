"""
Usage:
    zot hist <input>...
"""
# zotmer/commands/hist.py: one line `file<TAB>count<TAB>number of distinct k-mers` per histogram bin.
import sys

from zotmer_amd.library.container import KmerSet
from zotmer_amd.library.usage import Spec

_SPEC = Spec(rest="<input>")


def main(argv):
    for path in _SPEC.parse(argv[1:], __doc__)["<input>"]:
        with KmerSet(path, "r") as z:
            for f, c in sorted((int(f), c) for f, c in z.meta.get("hist", {}).items()):
                print("%s\t%d\t%d" % (path, f, c))


if __name__ == "__main__":
    main(["hist"] + sys.argv[1:])
