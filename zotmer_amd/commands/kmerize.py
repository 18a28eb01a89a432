"""
Usage:
    zot kmerize [options] <k> <output> <input>...

Kmerize FASTA or FASTQ inputs to produce a standard container object.

Arguments:
    <k>         the length of the k-mers (1..32). Recommended values: 10-30
    <output>    the name of the output file.
                recommended naming convention
                    - mykmers.k25 for a k-mer set of 25-mers
                    - mykmers.kf25 for a k-mer frequency set of 25-mers
                    - mykmers.e25 for an expanded k-mer set of 25-mers

Options:
    -m MEM      per-batch input size on the GPU (in MB); the result does not depend on it
    -C BAITS    capture mode - use kmers from the given FASTA file.
    -D FRAC     subsample k-mers, using FRAC proportion of k-mers
    -S SEED     if -D is given, give a seed for determining the
                subspace (defaults to 0).
    -k K        the k-mer length, as an alternative to the positional <k>
    -v          produce verbose progress messages
"""
# Drop-in for zotmer/commands/kmerize.py (reference file:line cited per step below).  The per-read
# Python loop, KmerAccumulator2, the spill files and mergeNinto are replaced by device batches:
# zk_kmerize (encode + radix sort + run-length count + strand mirror) and zk_union_sum.
import os
import sys
import time

import numpy as np

from zotmer_amd.library import engine, seqio, vectors
from zotmer_amd.library.container import KmerSet
from zotmer_amd.library.usage import Spec

_SPEC = Spec(options={"-m": True, "-C": True, "-D": True, "-S": True, "-k": True, "-v": False},
             positionals=[], rest="<args>", rest_min=2)


def main(argv):
    opts = _SPEC.parse(argv[1:], __doc__)
    args = opts["<args>"]
    if opts["-k"] is not None:                      # BASELINE.json spells it `zot kmerize -k 25 out in...`
        K, out, inputs = int(opts["-k"]), args[0], args[1:]
    else:                                           # the reference's positional form (kmerize.py:3,455)
        if len(args) < 3:
            _SPEC._die("wrong number of arguments", __doc__)
        K, out, inputs = int(args[0]), args[1], args[2:]
    if not inputs:
        _SPEC._die("no input files", __doc__)
    if not 1 <= K <= 32:
        raise SystemExit("zot kmerize: k must be between 1 and 32")
    verbose = opts["-v"]

    dist, world, rank = engine.distributed()        # one process per GPU under torch.distributed.run
    ctx = engine.context()
    subsample = None
    if opts["-D"] is not None:                      # kmerize.py:469-478
        subsample = (float(opts["-D"]), int(opts["-S"]) if opts["-S"] is not None else 0)
    baits = None
    if opts["-C"] is not None:                      # kmerize.py:480-485: both-strand k-mers of the bait FASTA
        acc = []
        for seq in seqio.fasta_sequences(opts["-C"]):
            d = ctx.upload_stream(seq + b"\n")
            ks, _ = ctx.encode(d, K, both=True)
            acc.append(ks.to_host())
        if acc:                                     # sorted distinct bait k-mers: device sort + run-length (zk_sort_count)
            allk = ctx.upload(np.concatenate(acc))
            baits = ctx.copy_of(ctx.sort_count(allk, 2 * K)[0])
        else:
            baits = ctx.upload(np.empty(0, np.uint64))

    table = engine.KmerTable(ctx, K, subsample=subsample, baits=baits)
    try:                                            # how much text is coming: lets the table memory be sized in one step
        total_in = sum(os.path.getsize(p) * (4 if p.endswith((".gz", ".bz2")) else 1) for p in inputs if os.path.isfile(p))
        if world == 1 and total_in:
            table.expect(total_in)
    except OSError:
        pass
    requested = (int(opts["-m"]) << 20) if opts["-m"] is not None else None
    n_reads = 0
    n_batch = 0                                     # multi-GPU: rank r counts batches r, r + world, ... (reads are independent)
    timing = os.environ.get("ZOT_TIMING") == "1"
    t_parse = t_gpu = 0.0
    t0 = time.perf_counter()
    for path in inputs:                             # files are simply processed in sequence (reads.py:66-93)
        native_fastq = (not seqio.is_fasta(path)) and path != "-" and not path.endswith(".bz2") and os.path.isfile(path)
        if native_fastq:
            # FASTQ, plain or gzip: the file is read ahead of the device by a native reader and parsed ON the device
            # (library/engine.py count_fastq_file, csrc/ingest.hip); the host never looks at the text
            size = os.path.getsize(path) * (4 if path.endswith(".gz") else 1)
            batch = engine.batch_bytes_for(ctx, requested, input_bytes=size)
            base = n_batch

            def take(i, base=base):
                return (base + i) % world == rank
            t1 = time.perf_counter()
            recs = engine.count_fastq_file(ctx, table, path, batch, take=take if world > 1 else None)
            n_batch += -(-size // batch)
            n_reads += recs
            t_gpu += time.perf_counter() - t1
            t0 = time.perf_counter()
        elif seqio.is_fasta(path):
            # FASTA records span lines and must be joined: host chunk parser, then upload
            batch = engine.batch_bytes_for(ctx, requested)
            for stream, recs in seqio.base_stream_batches([path], batch_bytes=batch):
                t1 = time.perf_counter()
                t_parse += t1 - t0
                if n_batch % world == rank:
                    table.add_stream(stream)
                n_batch += 1
                t0 = time.perf_counter()
                t_gpu += t0 - t1
                n_reads += recs                     # kmerize.py:527: every record counts
        else:
            # stdin / .bz2: Python reads (and decompresses) the text, the device still parses it (zk_fastq_mask)
            batch = engine.batch_bytes_for(ctx, requested)
            for text, phase, recs in seqio.fastq_text_batches(path, batch_bytes=batch):
                t1 = time.perf_counter()
                t_parse += t1 - t0
                if n_batch % world == rank:
                    table.add_fastq_text(text, phase)
                n_batch += 1
                t0 = time.perf_counter()
                t_gpu += t0 - t1
                n_reads += recs
        if verbose:
            sys.stderr.write("\r%d reads" % n_reads)
    if verbose:
        sys.stderr.write("\n")

    if dist is not None:
        kmers, counts, hist = _gather_distributed(ctx, dist, K, table)
        if rank != 0:
            return
    else:
        kmers, counts, hist = table.device_result()
    with KmerSet(out, "w") as z:                    # kmerize.py:541-561; delta + codec64 done on the device
        vectors.device_write_kmers_and_counts(ctx, z, kmers, counts)
        total = float(sum(table.acgt))
        z.meta["K"] = K
        z.meta["kmers"] = "kmers"
        z.meta["counts"] = "counts"
        z.meta["hist"] = hist
        z.meta["acgt"] = [c / total for c in table.acgt]    # ZeroDivisionError on empty input, as the reference
        z.meta["reads"] = n_reads
    if timing:
        sys.stderr.write("zot kmerize: read+parse %.2f s, upload+count %.2f s, hist+encode+write %.2f s\n"
                         % (t_parse, t_gpu, time.perf_counter() - t0))


def _gather_distributed(ctx, dist, K, table):
    """The ranks' tables meet in one exchange by k-mer owner (zotmer_amd/parallel.py): canonical lists when there is no -D
    (half the bytes; the strands are rebuilt on the owner), acgt is all-reduced, and rank 0 gathers the owned pieces to
    write the one output file."""
    from zotmer_amd import parallel
    comm = parallel.make_comm(ctx, dist)
    ex = parallel.Exchange(ctx, dist, K, owner=os.environ.get("ZOT_OWNER", "range"), comm=comm)
    if table.canonical:
        kt, ct, n = ex.ops.to_tensors(*table.canonical_result())
        k, c = ex.kmerize_finish(kt, ct, n)
    else:
        kmers, counts, _ = table.device_result()
        kt, ct, n = ex.ops.to_tensors(kmers, counts)
        if ex.owner == "range":
            ex.balanced_cuts([(kt, n)])
        k, c = ex.exchange_and_merge(kt, ct, n)
    table.acgt = comm.all_reduce(table.acgt)
    gk, gc = ex.gather_to_root(k, c)
    if comm.rank != 0:
        return None, None, None
    return gk, gc, ctx.hist(gc)


if __name__ == "__main__":
    main(["kmerize"] + sys.argv[1:])
