"""Build the gfx950 HIP library in-tree (hipcc cross-compiles without a GPU).

The library is rebuilt whenever the SHA-256 over every source it is compiled from (csrc/*.hip, csrc/*.hpp, the public
header) differs from the hash recorded next to it at build time; `_lib.load()` refuses a library whose recorded hash does
not match the tree it sits in (a stale .so would otherwise travel to the GPU box silently: it is git-ignored, not
gpurun-ignored)."""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmeshanything_amd.so")
HASH_FILE = OUT + ".srchash"
SOURCES = ["engine.hip"]
HEADER = os.path.normpath(os.path.join(HERE, "..", "include", "meshanything_amd.h"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unused-result"]


def source_files():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))) + [HEADER]


def source_hash() -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()


def recorded_hash() -> str:
    try:
        with open(HASH_FILE) as f:
            return f.read().strip()
    except OSError:
        return ""


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def needs_build() -> bool:
    return not os.path.exists(OUT) or recorded_hash() != source_hash()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    want = source_hash()
    cmd = [_hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT + ".tmp", "-ldl"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(OUT + ".tmp", OUT)
    with open(HASH_FILE, "w") as f:
        f.write(want + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force=True))
