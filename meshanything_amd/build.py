"""Build the gfx950 HIP library in-tree (hipcc cross-compiles without a GPU).

The library is rebuilt whenever the SHA-256 over every source it is compiled from (csrc/*.hip, csrc/*.hpp, the public
header, the flags -- NOT the compiler: a hipcc / ROCm upgrade needs `build(force=True)`) differs from the hash compiled INTO it (`ma_version()` ends in `src=<hash>`; a
copy is kept in a side file so that `needs_build()` does not have to dlopen).  `_lib.load()` refuses a library whose embedded
hash does not match the tree it sits in (a stale .so would otherwise travel to the GPU box silently: it is git-ignored, not
gpurun-ignored); a packaged library without the sources next to it is not checked, and MA_ALLOW_STALE_LIB=1 downgrades the
refusal to a warning."""
from __future__ import annotations

import ctypes
import glob
import hashlib
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["engine.hip"]
HEADER = os.path.normpath(os.path.join(HERE, "..", "include", "meshanything_amd.h"))
# MA_DEBUG=1 selects the debug variant ("race detection / sanitizers"): -O1 -g, device-side assert()s alive
# (the release build defines NDEBUG), its own file name so the two never shadow each other; `_lib.load()` follows the same variable.
# MA_DEBUG=asan additionally asks for HIP AddressSanitizer (host + device instrumentation; needs an xnack+ capable setup).
# MA_EXPERIMENTAL=1 compiles the measured-and-rejected decode-step forms in as well (csrc/experimental/: persist.hpp, rows_fused.hpp, layer_fused.hpp,
# the dense GEMM's A/B variants); the product build leaves them out.  Its own file name, like the debug variant.
EXPERIMENTAL = os.environ.get("MA_EXPERIMENTAL", "") not in ("", "0")
DEBUG = os.environ.get("MA_DEBUG", "") not in ("", "0")
ASAN = os.environ.get("MA_DEBUG", "") == "asan"
OUT = os.path.join(HERE, "libmeshanything_amd" + ("_debug" if DEBUG else "") + ("_exp" if EXPERIMENTAL else "") + ".so")
HASH_FILE = OUT + ".srchash"
_COMMON = ["-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unused-result"]
if ASAN:
    FLAGS = ["--offload-arch=gfx950:xnack+", "-O1", "-g", "-fsanitize=address", "-shared-libsan", "-DMA_DEBUG=1"] + _COMMON
elif DEBUG:
    FLAGS = ["--offload-arch=gfx950", "-O1", "-g", "-DMA_DEBUG=1"] + _COMMON
else:
    FLAGS = ["--offload-arch=gfx950", "-O3", "-DNDEBUG"] + _COMMON
if EXPERIMENTAL:
    FLAGS = FLAGS + ["-DMA_EXPERIMENTAL=1"]


def source_files():
    # csrc/experimental/: the measured-and-rejected decode-step forms (compiled only with MA_EXPERIMENTAL=1, but always part of the hash: one
    # rule for both library variants)
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "experimental", "*.hpp"))) + [HEADER]


def have_sources() -> bool:
    return all(os.path.exists(os.path.join(CSRC, f)) for f in SOURCES) and os.path.exists(HEADER)


def source_hash() -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()


def embedded_hash(path: str = OUT) -> str:
    """The source hash compiled into the library (`ma_version()`), "" if it cannot be read."""
    try:
        lib = ctypes.CDLL(path)
        lib.ma_version.restype = ctypes.c_char_p
        m = re.search(r"src=([0-9a-f]{64})", lib.ma_version().decode())
        return m.group(1) if m else ""
    except (OSError, AttributeError):
        return ""


def recorded_hash() -> str:
    try:
        with open(HASH_FILE) as f:
            return f.read().strip()
    except OSError:
        return embedded_hash() if os.path.exists(OUT) else ""


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
    cmd = [_hipcc()] + FLAGS + [f'-DMA_SRC_HASH="{want}"'] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT + ".tmp", "-ldl"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(OUT + ".tmp", OUT)
    with open(HASH_FILE, "w") as f:
        f.write(want + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force=True))
