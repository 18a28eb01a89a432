"""ZOT_TIMING=2: wall time of the phases of a command on stderr, one line per phase (a development aid; tools/bench_e2e.py sums
the lines by phase for the end-to-end record)."""
import os
import sys
import time

TIMING = os.environ.get("ZOT_TIMING") == "2"


class Phase:
    """with Phase(ctx, name[, bytes]): ... -- synchronises the context around the block when timing is on; free otherwise"""

    def __init__(self, ctx, what, nbytes=None):
        self.ctx, self.what, self.nbytes = ctx, what, nbytes

    def __enter__(self):
        if TIMING:
            self.ctx.sync()
            self.t = time.perf_counter()

    def __exit__(self, *a):
        if TIMING:
            self.ctx.sync()
            dt = time.perf_counter() - self.t
            rate = "  (%.1f GB/s)" % (self.nbytes / dt / 1e9) if self.nbytes and dt > 0 else ""
            sys.stderr.write("  [engine] %-28s %8.1f ms%s\n" % (self.what, dt * 1e3, rate))
