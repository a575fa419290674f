"""ctypes binding of libmeshanything_amd.so (C ABI: include/meshanything_amd.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the product path raises.
Build it with `python __graft_entry__.py` (or `meshanything_amd.build.build()`).
"""
from __future__ import annotations

import ctypes as C
import os

from .config import CMAConfig

HERE = os.path.dirname(os.path.abspath(__file__))
def _on(var: str) -> bool:
    return os.environ.get(var, "") not in ("", "0")


# build.py: MA_DEBUG selects the debug variant, MA_EXPERIMENTAL the one that also carries the rejected decode-step forms
LIB_PATH = os.path.join(HERE, "libmeshanything_amd" + ("_debug" if _on("MA_DEBUG") else "") + ("_exp" if _on("MA_EXPERIMENTAL") else "") + ".so")

MA_OK = 0
ERR_NAMES = {0: "MA_OK", -1: "MA_ERR_INVALID", -2: "MA_ERR_HIP", -3: "MA_ERR_STATE", -4: "MA_ERR_UNKNOWN_TENSOR",
             -5: "MA_ERR_SHAPE", -6: "MA_ERR_MISSING", -7: "MA_ERR_NCCL"}
DT_F32, DT_BF16, DT_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


class MAError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4), ("data", C.c_void_p)]


class SampleCfg(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("do_sample", C.c_int32), ("top_k", C.c_int32), ("top_p", C.c_float),
                ("max_new_tokens", C.c_int32), ("suppress_eos", C.c_int32), ("check_every", C.c_int32), ("logits_first_step", C.c_int32),
                ("seed", C.c_uint64), ("uniforms", C.c_void_p), ("forced_tokens", C.c_void_p), ("logits_out", C.c_void_p)]


class KernelTiming(C.Structure):
    _fields_ = [("launches", C.c_int32 * 8), ("ms", C.c_float * 8), ("step_ms_graph", C.c_float), ("step_ms_eager", C.c_float)]


# every symbol include/meshanything_amd.h declares: name -> (restype, argtypes)
_P, _I, _F = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    "ma_version": (C.c_char_p, []),
    "ma_last_error": (C.c_char_p, [_P]),
    "ma_engine_create": (_I, [C.POINTER(_P), C.POINTER(CMAConfig), _I]),
    "ma_engine_destroy": (None, [_P]),
    "ma_engine_set_option": (_I, [_P, C.c_char_p, C.c_int64]),
    "ma_engine_get_option": (_I, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "ma_engine_load_weights": (_I, [_P, C.POINTER(TensorDesc), _I]),
    "ma_engine_finalize_weights": (_I, [_P]),
    "ma_engine_arena": (_I, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "ma_engine_mark_weights_loaded": (_I, [_P]),
    "ma_engine_broadcast_weights": (_I, [_P, _P, _I, _P]),
    "ma_arena_bytes": (C.c_int64, [C.POINTER(CMAConfig)]),
    "ma_arena_num_entries": (_I, [C.POINTER(CMAConfig)]),
    "ma_arena_entry": (_I, [C.POINTER(CMAConfig), _I, C.c_char_p, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                            C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ma_pack_weights_host": (_I, [C.POINTER(CMAConfig), C.POINTER(TensorDesc), _I, _P, C.c_char_p, _I]),
    "ma_engine_upload_arena": (_I, [_P, _P, C.c_size_t]),
    "ma_encode": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "ma_to_shape_latents": (_I, [_P, _P, _I, _P, _P]),
    "ma_process_point_feature": (_I, [_P, _P, _I, _P, _P]),
    "ma_get_codes": (_I, [_P, _P, _I, _P, _P]),
    "ma_generate": (_I, [_P, _P, _I, C.POINTER(SampleCfg), _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    "ma_postprocess_tokens": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "ma_detokenize": (_I, [_P, _P, _P, _I, _P, _P]),
    "ma_detokenize_embeds": (_I, [_P, _P, _P, _P, _I, _P, _P]),
    "ma_forward": (_I, [_P, _P, _I, _I, C.POINTER(SampleCfg), _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P, _P, _P]),
    "ma_op_gemv": (_I, [_I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "ma_op_gemm": (_I, [_I, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "ma_op_gemm_bf16": (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "ma_op_layernorm": (_I, [_P, _I, _P, _P, _F, _P, _I, _I, _I, _P]),
    "ma_op_attention": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _F, _I, _I, _P]),
    "ma_op_decode_attention": (_I, [_I, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "ma_op_decode_attention_rows": (_I, [_P, _P, _P, _I, _I, _I, _I, C.c_size_t, _I, _I, _P, _P]),
    "ma_decode_attention_workspace_bytes": (C.c_size_t, [_I]),
    "ma_profile_decode": (_I, [_P, _I, _I, C.POINTER(KernelTiming), _P]),
    "ma_trace_decode": (_I, [_P, _I, _P, _I, _I, _P, _P, C.POINTER(C.c_int32), _P]),
    "ma_op_gemm_dec": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ma_op_gemm_dec_ln": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, C.c_float, _P, _P, _P, _I, _I, _I, _P]),
    "ma_op_gemm_dec_qkv": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, C.c_size_t, _P]),
    "ma_op_rows_prologue": (_I, [_I, _P, _I, _I, _P, _P, _P, _P, _F, _P, _I, _P, _P, _I, _P]),
    "ma_op_occupy_cus": (_I, [_I, _I, C.c_int64, _P, _P]),
    "ma_op_stream_copy": (_I, [_P, _P, C.c_size_t, _I, _P]),
    "ma_op_set_half_dtype": (_I, [_I]),
    "ma_engine_persist_available": (_I, [_P]),
    "ma_persist_trace": (_I, [_P, _I, _P, C.POINTER(C.c_int32), _P]),
    "ma_engine_read_logits": (_I, [_P, _I, _P, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it is missing -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: the MI355X HIP library has not been built "
                          f"(run `python __graft_entry__.py` / meshanything_amd.build.build()).  There is no CPU fallback.")
    from . import build as _build
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    if _build.have_sources():            # a packaged library without csrc/ next to it has nothing to be stale against
        have = _build.embedded_hash(LIB_PATH)
        if have != _build.source_hash():
            msg = (f"{LIB_PATH} was built from different sources than the ones next to it (embedded source hash {have[:12] or 'none'}, tree "
                   f"{_build.source_hash()[:12]}: csrc/*.hip, csrc/*.hpp or include/meshanything_amd.h changed since): rebuild with "
                   f"`python __graft_entry__.py`, or set MA_ALLOW_STALE_LIB=1 to load it anyway.")
            if os.environ.get("MA_ALLOW_STALE_LIB", "") != "1":
                raise ImportError(msg)
            import warnings
            warnings.warn(msg)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, engine=None) -> None:
    if rc != MA_OK:
        msg = load().ma_last_error(engine)
        raise MAError(rc, msg.decode() if msg else "")
