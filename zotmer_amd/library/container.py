"""
The on-disk sorted k-mer-set container, Python 3 side.

Layout (reference: zotmer/library/container/casket.py:16-21,219-234 and zotmer/library/kmers.py:8-21;
SURVEY.md appendix A):

    [member bytes ...][TOC as JSON][uint64 LE = len(TOC JSON)]

TOC = {"name": [[offset, length], ...]}; a name may occur more than once and readers take the last
entry.  A k-mer set has the members "kmers" (codec64 of the deltas of the ascending k-mers),
"counts" (codec64 of the counts) and "__meta__" (JSON: K, kmers, counts, hist, acgt, reads).
"""
import json
import os
import struct


class Container:
    """Append-only writer / random-access reader for one container file."""

    def __init__(self, path, mode="r"):
        if mode not in ("r", "w"):
            raise ValueError("mode must be 'r' or 'w'")
        self.path, self.mode = path, mode
        self.toc = {}
        self._streaming = False
        self._f = open(path, "rb" if mode == "r" else "wb")
        if mode == "r":
            self._load_toc()

    # -- reading ----------------------------------------------------------------------------
    def _load_toc(self):
        f = self._f
        f.seek(0, os.SEEK_END)
        size = f.tell()
        if size < 8:
            raise IOError("%s: not a k-mer container (too short)" % self.path)
        f.seek(size - 8)
        (n,) = struct.unpack("<Q", f.read(8))
        if n + 8 > size:
            raise IOError("%s: not a k-mer container (bad table of contents)" % self.path)
        f.seek(size - 8 - n)
        self.toc = json.loads(f.read(n).decode())

    def names(self):
        """[(member, length of its latest version)] sorted by name."""
        return sorted((nm, v[-1][1]) for nm, v in self.toc.items())

    def read(self, name):
        """The bytes of the latest version of a member (KeyError if absent)."""
        off, length = self.toc[name][-1]
        self._f.seek(off)
        data = self._f.read(length)
        if len(data) != length:
            raise IOError("%s: member %r is truncated" % (self.path, name))
        return data

    def member_size(self, name):
        return self.toc[name][-1][1]

    def read_range(self, name, start, length):
        """`length` bytes of the latest version of a member, from byte `start` of the member."""
        off, total = self.toc[name][-1]
        if start < 0 or length < 0 or start + length > total:
            raise ValueError("range outside member %r" % name)
        self._f.seek(off + start)
        data = self._f.read(length)
        if len(data) != length:
            raise IOError("%s: member %r is truncated" % (self.path, name))
        return data

    # -- writing ----------------------------------------------------------------------------
    def add(self, name, data):
        """Append a member given as bytes (or anything with the buffer protocol)."""
        if self.mode != "w":
            raise IOError("container opened read-only")
        if self._streaming:
            raise IOError("cannot add to the container while a streamed member is open")
        f = self._f
        f.seek(0, os.SEEK_END)
        off = f.tell()
        mv = memoryview(data).cast("B")
        f.write(mv)
        self.toc.setdefault(name, []).append([off, len(mv)])

    def add_device(self, name, ctx, arr):
        """Append a member straight from device memory (D2H through page-locked buffers, parallel pwrite: csrc/ingest.hip)."""
        if self.mode != "w":
            raise IOError("container opened read-only")
        if self._streaming:
            raise IOError("cannot add to the container while a streamed member is open")
        f = self._f
        f.seek(0, os.SEEK_END)
        off = f.tell()
        f.flush()
        if arr.nbytes:
            ctx.device_to_file(arr, f.fileno(), off)
            f.seek(0, os.SEEK_END)
        self.toc.setdefault(name, []).append([off, arr.nbytes])

    def read_device(self, name, ctx, dtype):
        """The latest version of a member as a device array of `dtype` (parallel pread, H2D through page-locked buffers)."""
        off, length = self.toc[name][-1]
        return ctx.file_to_device(self._f.fileno(), off, length, dtype)

    def add_stream(self, name):
        """A write()-able object; its bytes become member `name` when it is closed."""
        if self.mode != "w":
            raise IOError("container opened read-only")
        if self._streaming:
            raise IOError("cannot add to the container while a streamed member is open")
        return _StreamedMember(self, name)

    def close(self):
        if self._f is None:
            return
        if self.mode == "w":
            blob = json.dumps(self.toc).encode()
            self._f.seek(0, os.SEEK_END)
            self._f.write(blob)
            self._f.write(struct.pack("<Q", len(blob)))
            self._f.flush()
        self._f.close()
        self._f = None

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        elif self._f is not None:
            self._f.close()
            self._f = None
        return False


class _StreamedMember:
    def __init__(self, owner, name):
        self.owner, self.name = owner, name
        owner._streaming = True
        owner._f.seek(0, os.SEEK_END)
        self.off = owner._f.tell()
        self.length = 0

    def write(self, data):
        mv = memoryview(data).cast("B")
        self.owner._f.write(mv)
        self.length += len(mv)

    def close(self):
        if self.owner._streaming:
            self.owner.toc.setdefault(self.name, []).append([self.off, self.length])
            self.owner._streaming = False

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        return False


class KmerSet(Container):
    """A container with the `.meta` dictionary of a k-mer set (zotmer/library/kmers.py:8-21):
    loaded from member '__meta__' on open, written back on close."""

    def __init__(self, path, mode="r"):
        super().__init__(path, mode)
        self.meta = {}
        if mode == "r":
            self.meta = json.loads(self.read("__meta__").decode())

    def close(self):
        if self._f is not None and self.mode == "w":
            self.add("__meta__", json.dumps(self.meta).encode())
        super().close()
