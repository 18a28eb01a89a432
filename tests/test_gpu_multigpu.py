"""
Multi-GPU building blocks on ONE MI355X, through the C-ABI: the hash-range partition, the synthetic set generators,
64-bit-count checksums, the piecewise undelta, a 64-set merge (BASELINE config 4's fan-in), the product functions of
zotmer_amd/parallel.py at 8 logical ranks (tests/_logical_ranks_gpu.py) and both RCCL transports with one rank
(tests/_rccl_one_rank.py).  Bit-exact against the oracle / numpy throughout.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import zkoracle as zo
from zotmer_amd import native, parallel, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


def test_synth_sets_device_matches_numpy(ctx):
    for s in (0, 5, 63):
        a = synth.config4_set_args(s, 0.0005)
        k, c = ctx.synth_set(a["seed"], a["first"], a["count"], a["key_bits"], mul=a["mul"], add=a["add"], mod=a["mod"])
        wk = synth.set_keys(a["seed"], a["first"], a["count"], a["key_bits"], mul=a["mul"], add=a["add"], mod=a["mod"])
        assert np.array_equal(k.to_host(), wk)
        assert np.array_equal(c.to_host(), synth.set_counts(a["seed"], wk))
    c3 = synth.CONFIG3
    ka, _ = ctx.synth_set(c3["seed"], 0, 30000, c3["key_bits"], counts=False)
    kb, _ = ctx.synth_set(c3["seed"], 15000, 30000, c3["key_bits"], counts=False)
    wa, wb = synth.set_keys(c3["seed"], 0, 30000, c3["key_bits"]), synth.set_keys(c3["seed"], 15000, 30000, c3["key_bits"])
    assert np.array_equal(ka.to_host(), wa) and np.array_equal(kb.to_host(), wb)
    assert ctx.split(ka, kb) == zo.split(wa, wb) and ctx.split(ka, kb)[0] == 15000


@pytest.mark.parametrize("world", [1, 2, 3, 8, 32])
@pytest.mark.parametrize("n", [0, 1, 777, 2048, 2049, 150_001])
def test_hash_partition_vs_numpy(ctx, world, n):
    rng = np.random.default_rng(n * 37 + world)
    keys = np.unique(rng.integers(0, 1 << 50, size=n, dtype=np.uint64))
    n = len(keys)
    own = np.array([parallel.hash_owner(x, 9, world, zo.murmer) for x in keys], dtype=np.int64) if n else np.empty(0, np.int64)
    order = np.argsort(own, kind="stable")
    offs = [0] + [int(v) for v in np.cumsum(np.bincount(own, minlength=world))]
    for cdt in (None, np.uint32, np.uint64):
        cnts = rng.integers(1, 1000, size=n).astype(cdt) if cdt is not None else None
        dk = ctx.upload(keys)
        dc = ctx.upload(cnts) if cnts is not None else None
        ok, oc, got = ctx.hash_partition(dk, dc, world, seed=9)
        assert got == offs
        assert np.array_equal(ok.to_host(), keys[order])
        if cnts is not None:
            assert np.array_equal(oc.to_host(), cnts[order])
    for o in range(world):                         # every piece is still sorted
        piece = keys[order][offs[o]:offs[o + 1]]
        assert np.all(piece[1:] > piece[:-1])


def test_hash_partition_rejects_a_bad_world(ctx):
    dk = ctx.upload(np.arange(10, dtype=np.uint64))
    with pytest.raises(native.ZotkError):
        ctx.hash_partition(dk, None, 33)
    with pytest.raises(native.ZotkError):
        ctx.hash_partition(dk, None, 0)


def test_checksum_counts_and_undelta(ctx):
    rng = np.random.default_rng(2)
    k = np.unique(rng.integers(0, 1 << 62, size=50_000, dtype=np.uint64))
    m = (1 << 64) - 1
    for cdt in (np.uint32, np.uint64):
        c = rng.integers(1, 1 << 31, size=len(k)).astype(cdt)
        want = (int(c.astype(object).sum()) & m, sum(int(a) * int(b) for a, b in zip(k, c)) & m,
                sum(zo.murmer(int(a), 0) * int(b) for a, b in zip(k, c)) & m)
        assert ctx.checksum_counts(ctx.upload(k), ctx.upload(c)) == want
    # files.undelta continued from a base, in pieces (what every rank of a multi-GPU `zot dist` does with its words)
    deltas = np.diff(np.concatenate([[np.uint64(0)], k])).astype(np.uint64)
    cut = 12_345
    a = ctx.undelta(ctx.upload(deltas[:cut]), 0)
    b = ctx.undelta(ctx.upload(deltas[cut:]), 0)
    ctx.add_u64(b, int(a.to_host()[-1]))
    assert np.array_equal(np.concatenate([a.to_host(), b.to_host()]), k)
    assert np.array_equal(ctx.undelta(ctx.upload(deltas[cut:]), int(k[cut - 1])).to_host(), k[cut:])


def test_merge_64_sets_vs_oracle(ctx):
    """the fan-in of BASELINE config 4 (a 6-level tree inside zk_merge_n), 64-bit counts, odd sizes and an empty set"""
    sets = []
    for s in range(64):
        a = synth.config4_set_args(s, 0.0006)
        if s == 17:
            a["count"] = 0
        if s % 5 == 0:
            a["count"] += s * 7 + 1
        k = synth.set_keys(a["seed"], a["first"], a["count"], a["key_bits"], mul=a["mul"], add=a["add"], mod=a["mod"])
        sets.append((k, synth.set_counts(a["seed"], k)))
    zs, zc, zacgt = zo.merge_n(25, sets)
    k, c, acgt = ctx.merge_n([(ctx.upload(a), ctx.upload(b)) for a, b in sets])
    assert np.array_equal(k.to_host(), zs) and np.array_equal(c.to_host(), zc) and acgt == zacgt
    hv, hf = zo.hist(zc)
    assert ctx.hist(c) == {int(a): int(b) for a, b in zip(hv, hf)}
    sums = [0, 0, 0]
    for a, b in sets:
        for i, v in enumerate(ctx.checksum_counts(ctx.upload(a), ctx.upload(b))):
            sums[i] = (sums[i] + v) & ((1 << 64) - 1)
    assert ctx.checksum_counts(k, c) == tuple(sums)            # the size-independent check bench.py uses at full size


def test_kmer_table_merge_tree_matches_one_batch(ctx):
    """engine.KmerTable: many small batches (binary-counter merge tree) == one batch == the oracle"""
    from zotmer_amd.library import engine
    reads = synth.read_strings(synth.DEFAULT_SEED, 0, 7000, 150, genome=30000, sub_thr=synth.frac32(0.005), n_thr=synth.frac32(0.001))
    want = zo.kmerize(25, reads)
    t = engine.KmerTable(ctx, 25)
    for i in range(0, len(reads), 1000):
        t.add_stream(np.frombuffer(("".join(r + "\n" for r in reads[i:i + 1000])).encode(), dtype=np.uint8))
    k, c, h = t.result()
    assert np.array_equal(k, want["kmers"]) and np.array_equal(c, want["counts"]) and t.acgt == want["acgt"]
    hv, hf = zo.hist(want["counts"])
    assert h == {int(a): int(b) for a, b in zip(hv, hf)}


def test_kmer_table_grows_after_a_slab_swap_when_memory_is_short(ctx, monkeypatch):
    """The merge at the bottom of the stack swaps the two slabs, which leaves the scratch slab at the table slab's full size.  When
    the table then has to grow with live entries and the allocation fails (old pair + new pair + that scratch do not fit together),
    the scratch slab -- nothing in it is live between merges -- is given up and the allocation tried again.  The failure is injected
    (zk_alloc refuses while the scratch slab holds memory); the arrays stay the oracle's."""
    from zotmer_amd.library import engine
    engine.release_table_memory(ctx)
    reads = synth.read_strings(synth.DEFAULT_SEED + 3, 0, 9000, 150, genome=0)          # no repeats: the table grows with every batch
    want = zo.kmerize(25, reads)
    t = engine.KmerTable(ctx, 25)
    batches = [np.frombuffer(("".join(r + "\n" for r in reads[i:i + 1000])).encode(), dtype=np.uint8) for i in range(0, len(reads), 1000)]
    for b in batches[:2]:
        t.add_stream(b)          # two tables of one level: merged at the bottom of the stack -> the slabs swap
    assert t.scratch.E > 0
    refused, released = [], []
    real_empty, orig_release = ctx.empty, engine.Slab.release

    def empty(n, dtype):
        if t.scratch.E and n > 300000 and np.dtype(dtype) == np.uint64:          # a grow of the table slab while the scratch slab holds memory
            refused.append(n)
            raise native.ZotkError(-2, "injected: out of device memory")
        return real_empty(n, dtype)

    def release(self):
        released.append(self.E)
        orig_release(self)
    monkeypatch.setattr(engine.Slab, "release", release)
    monkeypatch.setattr(ctx, "empty", empty)
    for b in batches[2:]:
        t.add_stream(b)
    monkeypatch.setattr(ctx, "empty", real_empty)
    assert refused and released and released[0] > 0, "the scratch slab was never given up"
    k, c, h = t.result()
    assert np.array_equal(k, want["kmers"]) and np.array_equal(c, want["counts"]) and t.acgt == want["acgt"]
    engine.release_table_memory(ctx)


def _run(script, *args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(HERE, script)] + list(args), capture_output=True, text=True, timeout=timeout)
    return r


def test_merge_and_dist_over_8_logical_ranks_on_one_gpu():
    r = _run("_logical_ranks_gpu.py")
    assert r.returncode == 0 and r.stdout.count("LOGICAL-RANKS-OK") == 2, r.stdout[-6000:] + r.stderr[-2000:]


def test_rccl_transports_with_one_rank():
    r = _run("_rccl_one_rank.py")
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL-ONE-RANK ")]
    assert line, r.stdout[-2000:] + r.stderr[-6000:]
    rec = json.loads(line[-1][len("RCCL-ONE-RANK "):])
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "rccl_one_rank.json"), "w") as f:
            json.dump(rec, f, indent=1)
    assert r.returncode == 0, (rec, r.stderr[-3000:])
    assert rec["torch_chunked_verified"] and rec["native_verified"] and rec["native_allreduce"] and rec["chunked_gt_1GiB_intact"]
    # the RCCL calls themselves (ncclSend / ncclRecv to self in rounds, ncclAllReduce), not only the plumbing around them
    assert rec["selfloop_send_recv_cases"] == 24 and not rec["selfloop_send_recv_bad"] and rec["selfloop_allreduce"]
    assert rec["selfloop_exchange_range_verified"] and rec["selfloop_exchange_hash_verified"]


def test_commands_distributed_code_path_with_one_rank(tmp_path):
    """`zot kmerize | merge | dist` through their multi-GPU code (engine.distributed -> parallel.Exchange -> zk_comm_*) with ONE
    rank (ZOT_FORCE_DIST=1): the files and the stdout must equal what the plain single-GPU commands give."""
    from zotmer_amd import synth as sy
    fq = []
    for i in range(2):
        p = tmp_path / ("r%d.fastq" % i)
        p.write_text(sy.fastq_text(sy.DEFAULT_SEED, 3000 * i, 4000, 150, genome=50000, sub_thr=sy.frac32(0.005), n_thr=sy.frac32(0.0005)))
        fq.append(str(p))
    zot = os.path.join(ROOT, "zot")

    def run(env_extra, *args):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, zot] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-3000:]
        return r.stdout

    def members(path):
        from zotmer_amd.library.container import Container
        with Container(str(path), "r") as z:
            return {nm: z.read(nm) for nm, _ in z.names() if nm != "__meta__"}, json.loads(z.read("__meta__").decode())

    outs = {}
    for tag, env in (("plain", {}), ("dist", {"ZOT_FORCE_DIST": "1"}), ("dist_hash_torch", {"ZOT_FORCE_DIST": "1", "ZOT_OWNER": "hash", "ZOT_COMM": "torch"})):
        a, b, m = tmp_path / (tag + "_a.k25"), tmp_path / (tag + "_b.k25"), tmp_path / (tag + "_m.k25")
        run(env, "kmerize", 25, a, fq[0])
        run(env, "kmerize", 25, b, fq[1])
        run(env, "merge", m, a, b, a)
        d25 = run(env, "dist", "-M", "*.qual", 25, a, b).replace(tag + "_", "")
        d12 = run(env, "dist", "-M", "jaccard.qual", 12, a, b).replace(tag + "_", "")
        outs[tag] = (members(a), members(b), members(m), d25, d12)
    for tag in ("dist", "dist_hash_torch"):
        for i in range(3):
            assert outs[tag][i][0] == outs["plain"][i][0], (tag, i)            # codec64 streams byte for byte
            for key in ("K", "hist", "acgt", "reads"):
                assert outs[tag][i][1].get(key) == outs["plain"][i][1].get(key), (tag, i, key)
        assert outs[tag][3] == outs["plain"][3] and outs[tag][4] == outs["plain"][4]
