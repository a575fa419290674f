"""The C-ABI shared library builds for gfx950, loads without a GPU and exports exactly what include/meshanything_amd.h
declares; the host-only entry points (arena layout, host-side packing, error reporting) work on CPU."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from meshanything_amd import _lib, build                           # noqa: E402
from meshanything_amd.checkpoint import synthetic_items            # noqa: E402
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F32  # noqa: E402
from meshanything_amd import dp                                    # noqa: E402


@pytest.fixture(scope="module")
def lib():
    build.build(force=False, verbose=False)          # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "meshanything_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"MA_API\s+[\w\s\*]+?\b(ma_\w+)\s*\(", text))


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = _declared_symbols()
    assert len(declared) >= 28
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.ma_version()


def test_arena_layout_is_a_pure_function_of_the_config(lib):
    for cfg in (MAConfig.tiny(dtype=DTYPE_BF16), MAConfig.tiny(dtype=DTYPE_F32), MAConfig.full(dtype=DTYPE_BF16)):
        c = cfg.to_c()
        nbytes = lib.ma_arena_bytes(C.byref(c))
        n = lib.ma_arena_num_entries(C.byref(c))
        assert nbytes > 0 and n > 0
        end = 0
        name = C.create_string_buffer(256)
        off, sz = C.c_int64(), C.c_int64()
        dt, rows, cols = C.c_int32(), C.c_int32(), C.c_int32()
        for i in range(n):
            assert lib.ma_arena_entry(C.byref(c), i, name, 256, C.byref(off), C.byref(sz), C.byref(dt), C.byref(rows), C.byref(cols)) == 0
            assert off.value % 16 == 0 and off.value >= end          # 16-byte aligned, non-overlapping, ascending
            assert sz.value == rows.value * cols.value * (2 if dt.value == 1 else 4)
            end = off.value + sz.value
        assert end <= nbytes
    full = MAConfig.full(dtype=DTYPE_BF16, enc_exact=0).to_c()
    assert 1.15e9 < lib.ma_arena_bytes(C.byref(full)) < 1.30e9       # ~596 M parameters, matrices bf16, vectors/tables fp32
    full = MAConfig.full(dtype=DTYPE_BF16).to_c()                    # default: the point encoder's ~200 M matrix elements stay fp32 (enc_exact)
    assert 1.55e9 < lib.ma_arena_bytes(C.byref(full)) < 1.65e9


def test_host_packing_is_strict(lib):
    cfg = MAConfig.tiny(dtype=DTYPE_BF16)
    items = list(synthetic_items(cfg))
    a = dp.pack_host_arena(cfg, items)
    b = dp.pack_host_arena(cfg, reversed(items))                     # order of arrival does not matter
    assert np.array_equal(a, b) and a.any()
    with pytest.raises(_lib.MAError) as e:                           # strict=True: a missing tensor is an error ...
        dp.pack_host_arena(cfg, items[:-1])
    assert e.value.code == -6
    with pytest.raises(_lib.MAError) as e:                           # ... and so is an unknown key
        dp.pack_host_arena(cfg, items + [("not.a.key", np.zeros(3, np.float32))])
    assert e.value.code == -4
    name, arr = items[5]
    with pytest.raises(_lib.MAError) as e:                           # ... and a wrong shape
        dp.pack_host_arena(cfg, items[:5] + [(name, np.zeros((3, 3), np.float32))] + items[6:])
    assert e.value.code == -5


def test_bad_config_is_rejected_without_a_gpu(lib):
    c = MAConfig.tiny().to_c()
    c.struct_size = 12
    assert lib.ma_arena_bytes(C.byref(c)) == -1
    h = C.c_void_p()
    assert lib.ma_engine_create(C.byref(h), C.byref(c), 0) == -1 and not h.value
    assert b"struct_size" in lib.ma_last_error(None)


def test_product_path_fails_loudly_without_a_gpu(lib):
    """No CPU fallback anywhere in the product: without a GPU the engine and the reference-named facade refuse to construct."""
    import types
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from meshanything_amd.engine import Engine
    from meshanything_amd.model import MeshAnything
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(MAConfig.tiny())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        MeshAnything(types.SimpleNamespace(llm="facebook/opt-350m", codebook_size=8192, codebook_dim=1024, n_max_triangles=800), device=0)
    # the product package never imports the oracle
    import pathlib
    import re
    for f in pathlib.Path(REPO, "meshanything_amd").rglob("*.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle\b", f.read_text(), flags=re.M), f


def test_stale_library_is_detected(lib, tmp_path, monkeypatch):
    """build.py hashes every file the library is compiled from (csrc/*.hip, csrc/*.hpp -- gemm_decode.hpp included -- and
    the public header): an edit to any of them makes needs_build() true and makes _lib.load() refuse the stale .so."""
    import shutil
    from meshanything_amd import build as B
    names = {os.path.basename(f) for f in B.source_files()}
    assert {"engine.hip", "gemm_decode.hpp", "gemv.hpp", "attn_decode.hpp", "meshanything_amd.h"} <= names
    assert not B.needs_build()                                       # the fixture just loaded a matching library
    csrc = tmp_path / "csrc"
    shutil.copytree(B.CSRC, csrc)
    monkeypatch.setattr(B, "CSRC", str(csrc))
    assert not B.needs_build()                                       # same bytes elsewhere: same hash
    with open(csrc / "gemm_decode.hpp", "a") as f:
        f.write("// edited\n")
    assert B.needs_build()
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(ImportError, match="built from different sources"):
        _lib.load()
    # an explicit override downgrades the refusal to a warning
    monkeypatch.setenv("MA_ALLOW_STALE_LIB", "1")
    with pytest.warns(UserWarning, match="built from different sources"):
        assert _lib.load() is not None
    monkeypatch.delenv("MA_ALLOW_STALE_LIB")
    monkeypatch.setattr(_lib, "_lib", None)


def test_source_hash_travels_inside_the_library(lib, tmp_path, monkeypatch):
    """The hash is compiled into the .so (`ma_version()`): a library without its side file is still recognised as current, and a
    packaged library with no sources next to it loads without a check."""
    from meshanything_amd import build as B
    assert B.embedded_hash() == B.source_hash()
    assert ("src=" + B.source_hash()) in lib.ma_version().decode()
    monkeypatch.setattr(B, "HASH_FILE", str(tmp_path / "no_such_side_file"))
    assert B.recorded_hash() == B.source_hash() and not B.needs_build()
    monkeypatch.setattr(_lib, "_lib", None)
    assert _lib.load() is not None
    monkeypatch.setattr(B, "CSRC", str(tmp_path / "no_sources_here"))
    assert not B.have_sources()
    monkeypatch.setattr(_lib, "_lib", None)
    assert _lib.load() is not None
    monkeypatch.setattr(_lib, "_lib", None)
