#!/usr/bin/env python3
"""
Resource usage of every kernel in a built libzotk.so, read from the code objects themselves.

The .hip_fatbin section of the shared library holds one clang offload bundle per translation unit;
each bundle carries a gfx950 ELF whose NT_AMDGPU_METADATA note lists, per kernel, the registers,
the LDS, the scratch ("private segment") and the number of spilled SGPRs / VGPRs.  This module
extracts them with llvm-objcopy + llvm-readelf (no GPU needed) so that tests/test_kernel_resources.py
can hold every product kernel to "no scalar spills, scratch within the allow-list".

usage: kernel_resources.py [libzotk.so] [name-filter]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "zotmer_amd", "libzotk.so")

_FIELDS = {
    ".name": ("name", str), ".sgpr_count": ("sgpr", int), ".vgpr_count": ("vgpr", int), ".agpr_count": ("agpr", int),
    ".sgpr_spill_count": ("sgpr_spill", int), ".vgpr_spill_count": ("vgpr_spill", int),
    ".private_segment_fixed_size": ("scratch", int), ".group_segment_fixed_size": ("lds", int),
    ".max_flat_workgroup_size": ("max_wg", int), ".uses_dynamic_stack": ("dyn_stack", str),
}


def code_objects(lib_path):
    """The gfx950 ELF images inside the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        data = open(fat, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        b = m.start()
        n = struct.unpack_from("<Q", data, b + 24)[0]
        p = b + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                out.append(data[b + off: b + off + size])
    return out


def kernels(lib_path=DEFAULT_LIB):
    """{demangled kernel name: dict(sgpr, vgpr, sgpr_spill, vgpr_spill, scratch, lds, ...)}"""
    res = {}
    for elf in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in txt.splitlines():
            s = line.strip()
            if s.startswith("- .agpr_count") or s.startswith("- .args"):
                cur = {}
                s = s[2:]
            if cur is None:
                continue
            m = re.match(r"(\.[a-z_]+):\s*(.*)$", s)
            if m and m.group(1) in _FIELDS:
                key, conv = _FIELDS[m.group(1)]
                v = m.group(2).strip().strip("'\"")
                cur[key] = conv(v) if conv is str else int(v, 0)
            if s.startswith(".wavefront_size"):
                if "name" in cur:
                    res[cur["name"]] = cur
                cur = None
    names = list(res)
    if names:
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
        res = {d.replace("void ", "").replace("zk::", ""): v for d, v in zip(dem, res.values())}
    return res


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else DEFAULT_LIB
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    for k, v in sorted(kernels(lib).items()):
        if flt in k:
            print("%-100s vgpr %3d sgpr %3d lds %6d scratch %4d spill s%d v%d" % (
                k[:100], v.get("vgpr", -1), v.get("sgpr", -1), v.get("lds", -1), v.get("scratch", -1),
                v.get("sgpr_spill", -1), v.get("vgpr_spill", -1)))
