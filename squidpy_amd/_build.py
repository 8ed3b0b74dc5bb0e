"""Build recipe for ``libsqgr.so`` (hipcc, gfx950 only).  Used by ``__graft_entry__.build()`` and
``python -m squidpy_amd._build``; the library is built in-tree (``squidpy_amd/csrc/libsqgr.so``)."""

from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libsqgr.so")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_fingerprint() -> str:
    """sha256 (16 hex digits) over the kernel sources and the C ABI header: stamps profiles taken from a build, so that
    bench.py never prices one build's timings with another build's counters."""
    import hashlib

    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(root, "include", "sqgr.h")]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "..", "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB_PATH
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function"]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
