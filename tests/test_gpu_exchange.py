"""GPU half of the multi-GPU exchange on one device (see tests/_exchange_single_gpu.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from zotmer_amd import native

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_exchange_pieces_merge_on_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(HERE, "_exchange_single_gpu.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "EXCHANGE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_merge_n_32bit_counts_and_preallocated_output():
    from oracle import zkoracle as zo
    ctx = native.Context(0)
    rng = np.random.default_rng(4)
    sets = []
    for s in range(5):
        n = int(rng.integers(1000, 60000))
        x = np.sort(rng.choice(np.arange(1 << 17, dtype=np.uint64) << np.uint64(20), size=n, replace=False))
        sets.append((x, rng.integers(1, 9, size=n, dtype=np.uint32)))
    zs, zc, acgt = zo.merge_n(25, [(a, b.astype(np.uint64)) for a, b in sets])
    ok, oc = ctx.empty(sum(len(a) for a, _ in sets), np.uint64), ctx.empty(sum(len(a) for a, _ in sets), np.uint32)
    k, c, gacgt = ctx.merge_n([(ctx.upload(a), ctx.upload(b)) for a, b in sets], out=(ok, oc))
    assert np.array_equal(k.to_host(), zs) and np.array_equal(c.to_host().astype(np.uint64), zc) and gacgt == acgt
    ctx.close()
