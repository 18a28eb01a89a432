"""
Build-time fence for the two GPU faults of round 1 ("GPU faults this round"; DESIGN.md section 4,
"Register caps and spills").  Both faulting builds -- rle_kernel under __launch_bounds__(512, 8) (28 B of scratch) and
an experimental pass_pipe_kernel (100 B of scratch) -- were kernels whose scalar ballot masks / buffer descriptors are
spilled to VGPR lanes AND whose vector registers are spilled to scratch memory at the same time; no build that has
only one of the two has ever misbehaved.  This test reads the resource notes of every kernel in the built
libzotk.so (no GPU needed) and fails when

  * a kernel has both SGPR spills and a private (scratch) segment,
  * a kernel outside the allow-list has any scratch at all,
  * an allow-listed kernel exceeds its scratch budget, or uses a dynamic stack.

So a change of launch bounds, tile geometry or compiler flags that brings the combination back is caught here, on
the CPU, before anything is launched.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr   # noqa: E402

LIB = os.path.join(ROOT, "zotmer_amd", "libzotk.so")

# kernels that may use scratch: (substring of the demangled name, bytes per lane).  The (key, payload) variants of the
# one-tile radix pass keep 16 keys + 16 payloads + 16 ranks per lane and spill 1-12 VGPRs (no SGPR spills);
# they have run in every GPU test since round 1.
SCRATCH_ALLOW = [("pass_kernel<", 64)]


@pytest.fixture(scope="module")
def table():
    if not os.path.exists(LIB):
        pytest.skip("libzotk.so not built")
    if not os.path.exists(os.path.join(kr.LLVM, "llvm-readelf")):
        pytest.skip("llvm-readelf not available")
    t = kr.kernels(LIB)
    assert len(t) > 40, "could not read the kernel metadata of libzotk.so"
    return t


def test_every_kernel_reports_its_resources(table):
    for name, r in table.items():
        for f in ("vgpr", "sgpr", "scratch", "sgpr_spill", "vgpr_spill", "lds"):
            assert f in r, (name, f)
        assert r["vgpr"] <= 128 or "pass" not in name, (name, r)      # 512-thread workgroups need <= 128 VGPRs to launch 2 per CU
        assert r.get("dyn_stack", "false") == "false", name


def test_no_kernel_combines_scalar_spills_with_scratch(table):
    bad = {n: r for n, r in table.items() if r["sgpr_spill"] > 0 and r["scratch"] > 0}
    assert not bad, "kernels with SGPR spills AND scratch (the combination of both round-1 faults): %r" % bad


def test_scratch_only_on_the_allow_list(table):
    for name, r in table.items():
        if r["scratch"] == 0:
            assert r["vgpr_spill"] == 0, (name, r)
            continue
        budget = [b for s, b in SCRATCH_ALLOW if s in name]
        assert budget, "%s uses %d B of scratch and is not on the allow-list" % (name, r["scratch"])
        assert r["scratch"] <= budget[0], "%s: %d B of scratch, budget %d" % (name, r["scratch"], budget[0])


def test_the_two_kernels_that_faulted_are_scratch_free(table):
    hit = 0
    for name, r in table.items():
        if name.startswith("rle_kernel(") or name.startswith("rle_prefix_kernel(") or name.startswith("pass_pipe_kernel<"):
            hit += 1
            assert r["scratch"] == 0 and r["vgpr_spill"] == 0, (name, r)
    assert hit >= 3
