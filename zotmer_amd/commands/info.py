"""
Usage:
    zot info <input>...
"""
# zotmer/commands/info.py: print the metadata items of each container, sorted by key.
import sys

from zotmer_amd.library.container import KmerSet
from zotmer_amd.library.usage import Spec

_SPEC = Spec(rest="<input>")


def main(argv):
    for path in _SPEC.parse(argv[1:], __doc__)["<input>"]:
        with KmerSet(path, "r") as z:
            for k in sorted(z.meta):
                print(k, z.meta[k])


if __name__ == "__main__":
    main(["info"] + sys.argv[1:])
