#!/usr/bin/env python3
"""Scale / overflow check on the GPU box: 35 M reads of poly-A give ONE k-mer pair (A^25, T^25) with
4.41 G occurrences each -- more than a 32-bit count holds.  The reference keeps kmerize counts in
array('I') and raises (commands/kmerize.py:373-374); here zk_kmerize must report ZK_EOVERFLOW, and a
slightly smaller input (34 M reads: 4.28 G < 2^32) must succeed with the exact count."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zotmer_amd import native
ctx = native.Context(0)
L, K = 150, 25
for R, expect_overflow in ((34_000_000, False), (35_000_000, True)):
    one = np.frombuffer(b"A" * L + b"\n", dtype=np.uint8)
    host = np.tile(one, 1_000_000)
    d = ctx.empty(R * (L + 1), np.uint8)
    up = ctx.upload(host)
    for i in range(R // 1_000_000):
        ctx._check(ctx.lib.zk_copy(ctx.h, d.ptr + i * host.size, up.ptr, host.size))
    ctx.sync()
    t0 = time.perf_counter()
    try:
        k, c, st = ctx.kmerize(d, K, cap=1024)
        got = (k.to_host().tolist(), c.to_host().tolist())
        print(R, "ok in %.2f s:" % (time.perf_counter() - t0), got, "expected count", R * (L - K + 1))
        assert not expect_overflow and got == ([0, (1 << 50) - 1], [R * (L - K + 1)] * 2)
    except native.ZotkError as e:
        print(R, "raised in %.2f s:" % (time.perf_counter() - t0), e)
        assert expect_overflow and e.code == native.ZK_EOVERFLOW
    del d, up
print("STRESS-OK")
