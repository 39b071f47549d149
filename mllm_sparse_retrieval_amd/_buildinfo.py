"""What binary is this? Hashes that tie measured artefacts (profiles/*_counters.json) to the kernels they were measured on.

kernel_source_sha256(): every file of csrc/ that can change device code (*.hip, *.hpp, *.h), names and contents, in
name order. code_object_sha256(): the `.hip_fatbin` section of the built libmsr.so (the gfx950 code objects), if the
library is there. bench.py compares the stamps a counters file carries with these and reports `counters_stale`.
"""
from __future__ import annotations

import hashlib
import os
import struct
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmsr.so")


def kernel_source_sha256(csrc=CSRC):
    h = hashlib.sha256()
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".hpp", ".h")):
            h.update(fn.encode() + b"\0")
            with open(os.path.join(csrc, fn), "rb") as f:
                h.update(f.read())
            h.update(b"\0")
    return h.hexdigest()


def code_object_sha256(lib=LIB):
    """sha256 of the .hip_fatbin section (ELF64 little endian section walk); None when the library or section is absent."""
    try:
        with open(lib, "rb") as f:
            b = f.read()
        if b[:4] != b"\x7fELF" or b[4] != 2:
            return None
        shoff, = struct.unpack_from("<Q", b, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)

        def sec(i):
            name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", b, shoff + i * shentsize)
            return name, off, size

        _, stroff, strsize = sec(shstrndx)
        names = b[stroff:stroff + strsize]
        for i in range(shnum):
            name, off, size = sec(i)
            if names[name:names.index(b"\0", name)] == b".hip_fatbin":
                return hashlib.sha256(b[off:off + size]).hexdigest()
    except Exception:
        return None
    return None


def git_head(root=os.path.dirname(PKG)):
    try:
        return subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              timeout=10).stdout.decode().strip() or None
    except Exception:
        return None


def stamp():
    return {"git_head": git_head(), "kernel_source_sha256": kernel_source_sha256(), "code_object_sha256": code_object_sha256()}


def stale_reason(file_stamp):
    """None when a counters file's stamp matches this tree's kernels, else why it does not."""
    if not isinstance(file_stamp, dict) or not file_stamp.get("kernel_source_sha256"):
        return "the counters file carries no stamp"
    if file_stamp["kernel_source_sha256"] != kernel_source_sha256():
        return "csrc/ kernel sources changed since the counters were collected"
    now = code_object_sha256()
    if now and file_stamp.get("code_object_sha256") and now != file_stamp["code_object_sha256"]:
        return "libmsr.so's gfx950 code objects differ from the profiled build"
    return None
