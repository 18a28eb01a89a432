"""
Sequence text -> base-stream batches for the GPU.

Reference: zotmer/library/reads.py:11-33,86-98 (file type by name suffix, FASTA else FASTQ) and
zotmer/library/file.py:19-52,79-123 (line parsers; .gz / .bz2 inputs are decompressed on the fly --
the reference pipes them through `gunzip -c` / `bunzip2 -c`, here Python's gzip / bz2 modules do it).

A base stream is every sequence of the batch followed by '\\n' (see include/zotk.h).  Parsing is done
by the native chunk parsers in libzotk.so (csrc/hostio.cpp); this module only moves bytes.
"""
import bz2
import ctypes as C
import gzip
import sys

import numpy as np

from zotmer_amd import native

_COMPRESSION = (".gz", ".bz2")
_FASTA = (".fa", ".fasta", ".fas", ".fna")


def strip_compression_suffix(name):
    for s in _COMPRESSION:
        if name.endswith(s):
            return name[:-len(s)]
    return name


def is_fasta(name):
    return strip_compression_suffix(name).endswith(_FASTA)


def open_binary(name):
    if name == "-":
        return sys.stdin.buffer
    if name.endswith(".gz"):
        return gzip.open(name, "rb")
    if name.endswith(".bz2"):
        return bz2.open(name, "rb")
    return open(name, "rb")


def base_stream_batches(paths, batch_bytes=256 << 20, chunk_bytes=32 << 20):
    """Yield (uint8 array holding a base stream, number of records in it) over all input files, in
    order; a batch never splits a record."""
    lib = native.load()
    out = np.empty(batch_bytes + chunk_bytes + 16, dtype=np.uint8)
    out_len = C.c_uint64(0)
    recs_in_batch = 0
    for path in paths:
        parse = lib.zk_parse_fasta if is_fasta(path) else lib.zk_parse_fastq
        state = (C.c_uint64 * 4)(0, 0, 0, 0)
        carry = b""
        with open_binary(path) as f:
            final = False
            while not final:
                data = f.read(chunk_bytes)
                final = len(data) < chunk_bytes
                if not final:
                    # peek: an exact multiple of the chunk size still needs a final call
                    pass
                buf = carry + data
                if not data:
                    final = True
                pos = 0
                while True:
                    if out_len.value + (len(buf) - pos) + 2 > out.size:
                        # flush what we have, then continue with the same text
                        if out_len.value:
                            yield out[:out_len.value].copy(), recs_in_batch
                            out_len.value = 0
                            recs_in_batch = 0
                        if (len(buf) - pos) + 2 > out.size:
                            out = np.empty(len(buf) - pos + chunk_bytes, dtype=np.uint8)
                    before = state[1]
                    consumed = C.c_uint64(0)
                    view = memoryview(buf)[pos:]
                    cbuf = (C.c_char * len(view)).from_buffer_copy(view) if len(view) else None
                    rc = parse(cbuf, len(view), int(final), state, out.ctypes.data, out.size, C.byref(out_len), C.byref(consumed))
                    recs_in_batch += state[1] - before
                    pos += consumed.value
                    if rc == native.ZK_OK:
                        break
                    if rc != native.ZK_ENOSPC:
                        raise IOError("parse error %d in %s" % (rc, path))
                carry = buf[pos:]
                if out_len.value >= batch_bytes:
                    yield out[:out_len.value].copy(), recs_in_batch
                    out_len.value = 0
                    recs_in_batch = 0
    if out_len.value or recs_in_batch:
        yield out[:out_len.value].copy(), recs_in_batch


def fasta_sequences(path):
    """The sequences of a (small) FASTA file as bytes objects -- used for bait files."""
    seqs = []
    for stream, _ in base_stream_batches([path]):
        seqs.extend(bytes(stream).split(b"\n")[:-1])
    return seqs


def fasta_records(path):
    """(name, sequence) per FASTA record, as file.readFasta yields them (library/file.py:19-36): lines are
    stripped and joined, the name is the header without '>' and surrounding blanks, text before the
    first header is dropped.  For the small per-record inputs of `zot jaccard`."""
    name, parts = None, []
    with open_binary(path) as f:
        for line in f:
            line = line.strip()
            if line[:1] == b">":
                if name is not None:
                    yield name, b"".join(parts)
                name, parts = line[1:].strip().decode("latin-1"), []
            else:
                parts.append(line)
    if name is not None:
        yield name, b"".join(parts)


def fastq_text_batches(path, batch_bytes=1 << 30):
    """Raw FASTQ text in batches that end at line ends, for the device-side parser (zk_fastq_mask):
    yields (bytes, line phase at the start of the batch, records completed inside the batch).  A trailing
    group of fewer than four lines is dropped, as file.readFastq does (library/file.py:51-52)."""
    lines = 0
    carry = b""
    with open_binary(path) as f:
        while True:
            data = f.read(batch_bytes)
            last = len(data) < batch_bytes
            buf = carry + data
            if not last:
                cut = buf.rfind(b"\n") + 1
                if cut == 0:                      # no line end at all in this much text: keep reading
                    carry = buf
                    continue
                carry, buf = buf[cut:], buf[:cut]
            else:
                carry = b""
                if buf and not buf.endswith(b"\n"):
                    buf += b"\n"                  # the last line counts even without a terminator
            nl = buf.count(b"\n")
            if last:
                extra = (lines + nl) % 4          # lines of an incomplete final record
                for _ in range(extra):
                    buf = buf[:buf.rfind(b"\n", 0, len(buf) - 1) + 1]
                nl -= extra
            if buf:
                before = lines // 4
                yield buf, lines % 4, (lines + nl) // 4 - before
                lines += nl
            if last:
                return
