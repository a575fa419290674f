"""Python handle on the HIP engine.  torch is used for device memory and streams only (plumbing): every
computation below happens inside libmeshanything_amd.so."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .config import MAConfig, DTYPE_BF16, DTYPE_F32


def _stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


class _ArenaView:
    """Exposes the engine's device weight arena through __cuda_array_interface__ so torch can alias it."""
    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class Engine:
    def __init__(self, cfg: MAConfig, device: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("meshanything_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self._c = cfg.to_c()
        h = C.c_void_p()
        _lib.check(self.lib.ma_engine_create(C.byref(h), C.byref(self._c), device))
        self.h = h
        self._keep = []

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.ma_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        _lib.check(rc, self.h)

    def set_option(self, name: str, value: int) -> None:
        self._check(self.lib.ma_engine_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int64()
        self._check(self.lib.ma_engine_get_option(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    # ---------------------------------------------------------------- weights (main.py:99-104)
    @staticmethod
    def _desc(name: str, arr) -> Tuple[_lib.TensorDesc, object]:
        if isinstance(arr, torch.Tensor):
            t = arr.detach().cpu().contiguous()
            if t.dtype == torch.bfloat16:
                a, dt = t.view(torch.int16).numpy(), _lib.DT_BF16
            elif t.dtype == torch.float16:
                a, dt = t.numpy(), _lib.DT_F16
            else:
                a, dt = t.float().numpy(), _lib.DT_F32
        else:
            a = np.ascontiguousarray(arr)
            if a.dtype == np.float16:
                dt = _lib.DT_F16
            else:
                a = np.ascontiguousarray(a, dtype=np.float32)
                dt = _lib.DT_F32
        shape = tuple(arr.shape)
        d = _lib.TensorDesc()
        d.name = name.encode()
        d.dtype = dt
        d.ndim = len(shape)
        for i, s in enumerate(shape):
            d.shape[i] = s
        d.data = a.ctypes.data
        return d, a

    def load_weights(self, items: Iterable[Tuple[str, object]], finalize: bool = True) -> None:
        """items: (reference state-dict key, ndarray | torch tensor).  Streams tensor by tensor (2.4 GB fp32 checkpoint)."""
        for name, arr in items:
            d, keep = self._desc(name, arr)
            self._check(self.lib.ma_engine_load_weights(self.h, C.byref(d), 1))
            del keep
        if finalize:
            self._check(self.lib.ma_engine_finalize_weights(self.h))

    def arena_tensor(self) -> torch.Tensor:
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.ma_engine_arena(self.h, C.byref(p), C.byref(n)))
        return torch.as_tensor(_ArenaView(p.value, n.value), device=self.device)

    def mark_weights_loaded(self) -> None:
        self._check(self.lib.ma_engine_mark_weights_loaded(self.h))

    def upload_arena(self, host: np.ndarray) -> None:
        self._check(self.lib.ma_engine_upload_arena(self.h, C.c_void_p(host.ctypes.data), host.nbytes))

    # ---------------------------------------------------------------- hot path
    def encode(self, pc_normal: torch.Tensor, want_prefix: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        cfg = self.cfg
        assert pc_normal.dim() == 3 and pc_normal.shape[1] == cfg.n_points and pc_normal.shape[2] == 6, pc_normal.shape
        x = pc_normal.to(self.device)
        if x.dtype == torch.float16:
            dt = _lib.DT_F16
        else:
            x, dt = x.float(), _lib.DT_F32
        x = x.contiguous()
        B = x.shape[0]
        latents = torch.empty(B, cfg.cond_length, cfg.enc_width, dtype=torch.float32, device=self.device)
        prefix = torch.empty(B, cfg.cond_length, cfg.hidden, dtype=torch.float32, device=self.device) if want_prefix else None
        self._check(self.lib.ma_encode(self.h, _ptr(x), dt, B, _ptr(latents), _ptr(prefix), _stream_ptr()))
        return latents, prefix

    def to_shape_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """point_encoder.to_shape_latents (asl_pl_module.py:182-185): (B, 256, 768) -> (B, 256, 768)."""
        cfg = self.cfg
        x = latents.to(self.device, torch.float32).contiguous()
        assert x.dim() == 3 and x.shape[1] == cfg.num_latents and x.shape[2] == cfg.enc_width, x.shape
        out = torch.empty_like(x)
        self._check(self.lib.ma_to_shape_latents(self.h, _ptr(x), x.shape[0], _ptr(out), _stream_ptr()))
        return out

    def process_point_feature(self, point_feature: torch.Tensor) -> torch.Tensor:
        """MeshAnything.process_point_feature (meshanything.py:125-132): (B, 257, 768) -> (B, 257, 1024)."""
        cfg = self.cfg
        x = point_feature.to(self.device, torch.float32).contiguous()
        assert x.dim() == 3 and x.shape[1] == cfg.cond_length and x.shape[2] == cfg.enc_width, x.shape
        out = torch.empty(x.shape[0], cfg.cond_length, cfg.hidden, dtype=torch.float32, device=self.device)
        self._check(self.lib.ma_process_point_feature(self.h, _ptr(x), x.shape[0], _ptr(out), _stream_ptr()))
        return out

    def get_codes(self, ids: torch.Tensor) -> torch.Tensor:
        """MeshAnything.get_codes (meshanything.py:178-212): ids (B, 9F) -> (B, 3F, codebook_dim)."""
        cfg = self.cfg
        ids = ids.to(self.device, torch.int64).contiguous()
        assert ids.dim() == 2 and ids.shape[1] == cfg.n_max_faces * 9, ids.shape
        out = torch.empty(ids.shape[0], cfg.n_max_faces * 3, cfg.codebook_dim, dtype=torch.float32, device=self.device)
        self._check(self.lib.ma_get_codes(self.h, _ptr(ids), ids.shape[0], _ptr(out), _stream_ptr()))
        return out

    def _sample_cfg(self, sampling: bool, max_new_tokens: Optional[int], suppress_eos: bool, uniforms: Optional[torch.Tensor],
                    seed: int, check_every: int, top_k: int, top_p: float, forced_tokens: Optional[torch.Tensor] = None,
                    logits_out: Optional[torch.Tensor] = None, logits_first_step: int = 0) -> Tuple[_lib.SampleCfg, object]:
        sc = _lib.SampleCfg()
        sc.struct_size = C.sizeof(_lib.SampleCfg)
        sc.do_sample = 1 if sampling else 0
        sc.top_k, sc.top_p = top_k, top_p
        sc.max_new_tokens = int(max_new_tokens or 0)
        sc.suppress_eos = 1 if suppress_eos else 0
        sc.check_every = check_every
        sc.seed = seed
        keep = []
        if uniforms is not None:
            keep.append(uniforms.to(self.device, torch.float32).contiguous())
            sc.uniforms = keep[-1].data_ptr()
        if forced_tokens is not None:
            keep.append(forced_tokens.to(self.device, torch.int64).contiguous())
            sc.forced_tokens = keep[-1].data_ptr()
        if logits_out is not None:
            sc.logits_out = logits_out.data_ptr()
            sc.logits_first_step = int(logits_first_step)
        return sc, keep

    def generate(self, prefix: torch.Tensor, sampling: bool = False, max_new_tokens: Optional[int] = None,
                 suppress_eos: bool = False, uniforms: Optional[torch.Tensor] = None, seed: int = 0, check_every: int = 64,
                 top_k: int = 50, top_p: float = 0.95, forced_tokens: Optional[torch.Tensor] = None, return_logits: bool = False,
                 logits_first_step: int = 0):
        """transformer.generate(inputs_embeds=prefix, ...) (meshanything.py:143-162) -> (tokens (B, n_generated), lengths).
        forced_tokens (B, max_new_tokens): teacher forcing -- the returned tokens are the engine's own picks at every step of the GIVEN
        stream (ma_sample_cfg.forced_tokens).  return_logits: also returns the (B, max_new_tokens - logits_first_step, vocab) fp32 logits of
        every step from `logits_first_step` on."""
        cfg = self.cfg
        prefix = prefix.to(self.device, torch.float32).contiguous()
        B = prefix.shape[0]
        maxn = int(max_new_tokens or cfg.max_new_tokens)
        if uniforms is not None:
            assert tuple(uniforms.shape) == (B, maxn), (uniforms.shape, (B, maxn))
        if forced_tokens is not None:
            assert tuple(forced_tokens.shape) == (B, maxn), (forced_tokens.shape, (B, maxn))
            lo, hi = int(forced_tokens.min()), int(forced_tokens.max())
            if lo < 0 or hi >= cfg.vocab:
                raise ValueError(f"forced_tokens must lie in [0, {cfg.vocab}), got [{lo}, {hi}]")
        assert 0 <= logits_first_step < maxn
        logits = torch.zeros(B, maxn - logits_first_step, cfg.vocab, dtype=torch.float32, device=self.device) if return_logits else None
        sc, keep = self._sample_cfg(sampling, max_new_tokens, suppress_eos, uniforms, seed, check_every, top_k, top_p, forced_tokens, logits,
                                    logits_first_step)
        tokens = torch.empty(B, cfg.max_new_tokens, dtype=torch.int64, device=self.device)
        lengths = (C.c_int32 * B)()
        ngen = C.c_int32()
        self._check(self.lib.ma_generate(self.h, _ptr(prefix), B, C.byref(sc), _ptr(tokens), lengths, C.byref(ngen), _stream_ptr()))
        del keep
        self._warn_if_fell_back()
        if return_logits:
            return tokens[:, :ngen.value], np.array(list(lengths), dtype=np.int32), logits[:, :max(0, ngen.value - logits_first_step)]
        return tokens[:, :ngen.value], np.array(list(lengths), dtype=np.int32)

    def postprocess_tokens(self, results: torch.Tensor) -> torch.Tensor:
        """meshanything.py:163-172.  results (B, n<=max_new) -> ids (B, 9*n_max_faces)."""
        cfg = self.cfg
        assert results.dim() == 2, results.shape
        B, n = results.shape
        if n > cfg.max_new_tokens:
            raise ValueError(f"{n} generated tokens per row, but 9*n_max_faces+2 = {cfg.max_new_tokens} is the most generate() can return")
        t = results.to(self.device, torch.int64)
        if n == 0:                                                   # nothing generated: every slot is the eos padding of meshanything.py:141-142
            t = torch.zeros(B, 1, dtype=torch.int64, device=self.device)     # (an empty tensor has no data pointer to hand over)
        if n > 0 and t.stride(1) != 1:
            t = t.contiguous()
        ld = t.stride(0) if (B > 1 and n > 0) else max(n, 1)        # rows are read in place (a [:, :n] view of generate()'s buffer)
        ids = torch.empty(B, cfg.n_max_faces * 9, dtype=torch.int64, device=self.device)
        self._check(self.lib.ma_postprocess_tokens(self.h, _ptr(t), ld, B, n, _ptr(ids), _stream_ptr()))
        return ids

    def detokenize(self, ids: torch.Tensor, latents: torch.Tensor, codes: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tokenizer(ids, codes, point_feature=latents) (meshanything.py:50-80, 173-174) -> (B, F, 3, 3).  codes = the
        reference's `input_embeds` (B, 3F, codebook_dim); None: get_codes(ids) is computed inside the launch chain."""
        cfg = self.cfg
        ids = ids.to(self.device, torch.int64)
        ids = ids.reshape(ids.shape[0], -1).contiguous()            # the reference reshapes too (meshanything.py:51)
        latents = latents.to(self.device, torch.float32).contiguous()
        B = ids.shape[0]
        if tuple(ids.shape) != (B, cfg.n_max_faces * 9):
            raise ValueError(f"ids must be (B, 9*n_max_faces = {cfg.n_max_faces * 9}), got {tuple(ids.shape)}")
        if tuple(latents.shape) != (B, cfg.cond_length, cfg.enc_width):
            raise ValueError(f"point_feature must be ({B}, {cfg.cond_length}, {cfg.enc_width}), got {tuple(latents.shape)}")
        if codes is not None:
            codes = codes.to(self.device, torch.float32).contiguous()
            if tuple(codes.shape) != (B, cfg.n_max_faces * 3, cfg.codebook_dim):
                raise ValueError(f"input_embeds must be ({B}, {cfg.n_max_faces * 3}, {cfg.codebook_dim}), got {tuple(codes.shape)}")
        coords = torch.empty(B, cfg.n_max_faces, 3, 3, dtype=torch.float32, device=self.device)
        self._check(self.lib.ma_detokenize_embeds(self.h, _ptr(ids), _ptr(codes), _ptr(latents), B, _ptr(coords), _stream_ptr()))
        return coords

    def forward(self, pc_normal: torch.Tensor, sampling: bool = False, max_new_tokens: Optional[int] = None,
                suppress_eos: bool = False, uniforms: Optional[torch.Tensor] = None, seed: int = 0, check_every: int = 64,
                top_k: int = 50, top_p: float = 0.95) -> Dict[str, object]:
        """MeshAnything.forward (meshanything.py:134-176) in one library call."""
        cfg = self.cfg
        x = pc_normal.to(self.device)
        if x.dtype == torch.float16:
            dt = _lib.DT_F16
        else:
            x, dt = x.float(), _lib.DT_F32
        x = x.contiguous()
        B = x.shape[0]
        sc, keep = self._sample_cfg(sampling, max_new_tokens, suppress_eos, uniforms, seed, check_every, top_k, top_p)
        coords = torch.empty(B, cfg.n_max_faces, 3, 3, dtype=torch.float32, device=self.device)
        tokens = torch.empty(B, cfg.max_new_tokens, dtype=torch.int64, device=self.device)
        ids = torch.empty(B, cfg.n_max_faces * 9, dtype=torch.int64, device=self.device)
        latents = torch.empty(B, cfg.cond_length, cfg.enc_width, dtype=torch.float32, device=self.device)
        lengths = (C.c_int32 * B)()
        ngen = C.c_int32()
        self._check(self.lib.ma_forward(self.h, _ptr(x), dt, B, C.byref(sc), _ptr(coords), _ptr(tokens), lengths, C.byref(ngen),
                                        _ptr(ids), _ptr(latents), _stream_ptr()))
        del keep
        self._warn_if_fell_back()
        return {"coords": coords, "tokens": tokens[:, :ngen.value], "lengths": np.array(list(lengths), dtype=np.int32),
                "ids": ids, "latents": latents}

    # ---------------------------------------------------------------- persistent decode step (csrc/experimental/persist.hpp)
    def persist_available(self) -> bool:
        """True when `set_option("decode_impl", 1)` can take effect: bf16, 350M layer shape, 256-CU device."""
        return bool(self.lib.ma_engine_persist_available(self.h))

    def read_logits(self, row: int = 0) -> torch.Tensor:
        """Logits (codebook_size + 3) of the most recent decode step of batch row `row`."""
        out = torch.empty(self.cfg.vocab, dtype=torch.float32, device=self.device)
        self._check(self.lib.ma_engine_read_logits(self.h, row, _ptr(out), _stream_ptr()))
        return out

    def persist_trace(self, kv_len: int) -> np.ndarray:
        """(256 workgroups, n_events) 100 MHz ticks of one persistent step: start, per edge {sweep start, gather done}, end."""
        out = np.zeros(256 * (320 + 512), dtype=np.uint64)
        n = C.c_int32()
        self._check(self.lib.ma_persist_trace(self.h, kv_len, C.c_void_p(out.ctypes.data), C.byref(n), _stream_ptr()))
        self.last_compute_trace = out[256 * 320:].reshape(256, 512)
        return out[:256 * 320].reshape(256, 320)[:, :n.value]

    def occupy_cus(self, n_blocks: int, microseconds: int, stream: Optional[torch.cuda.Stream] = None, lds_bytes: int = 160 * 1024,
                   release: Optional[torch.Tensor] = None) -> None:
        """Test aid (ma_op_occupy_cus): park `n_blocks` workgroups holding `lds_bytes` of LDS each (default: a whole CU) on `stream` for
        `microseconds` -- or until `release` (a pinned int32 host tensor the caller sets to non-zero) says so -- so that a test can take
        CUs away from the engine's own stream."""
        sp = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        if release is not None:
            assert release.dtype == torch.int32 and release.is_pinned()
        self._check(self.lib.ma_op_occupy_cus(int(n_blocks), int(lds_bytes), int(microseconds), _ptr(release), sp))

    # ---------------------------------------------------------------- measurement
    def profile_decode(self, kv_len: int, steps: int = 4) -> Dict[str, object]:
        kt = _lib.KernelTiming()
        self._check(self.lib.ma_profile_decode(self.h, kv_len, steps, C.byref(kt), _stream_ptr()))
        names = ["gemv", "attn_decode", "persist", "pick"]
        return {"launches": {n: kt.launches[i] for i, n in enumerate(names)}, "ms": {n: kt.ms[i] for i, n in enumerate(names)},
                "step_ms_graph": kt.step_ms_graph, "step_ms_eager": kt.step_ms_eager, "steps": steps, "kv_len": kv_len}


    def _warn_if_fell_back(self) -> None:
        """One line when a generation lost its fused decode launches (an in-launch exchange timed out: another tenant held the CUs) and ran
        on the five-launch chain instead -- same tokens, a slower step; the engine re-arms the fused launches after 16 clean generations."""
        n = self.get_option("chain_fallbacks")
        seen = getattr(self, "_fallbacks_seen", 0)
        if n > seen:
            import warnings
            warnings.warn(f"meshanything_amd: the fused decode launches timed out {n - seen} time(s) (shared device?); the generation re-ran on "
                          f"the launch chain (same results, slower decode step); chain_fallbacks = {n}, error word {self.get_option('xchg_last_code')}", RuntimeWarning, stacklevel=3)
            self._fallbacks_seen = n

    def trace_decode(self, kv_len: int, max_launches: int = 160, max_blocks: int = 2560) -> Dict[str, np.ndarray]:
        """In-kernel timeline of one eager decode step (diagnostics): ticks of the 100 MHz real-time counter."""
        out = np.zeros((max_launches, max_blocks, 4), dtype=np.uint64)
        kinds = np.zeros(max_launches, dtype=np.int32)
        blocks = np.zeros(max_launches, dtype=np.int32)
        n = C.c_int32()
        self._check(self.lib.ma_trace_decode(self.h, kv_len, C.c_void_p(out.ctypes.data), max_launches, max_blocks,
                                             C.c_void_p(kinds.ctypes.data), C.c_void_p(blocks.ctypes.data), C.byref(n), _stream_ptr()))
        return {"ticks": out[:n.value], "kinds": kinds[:n.value], "blocks": blocks[:n.value]}


def build_engine(cfg: MAConfig, items: Iterable[Tuple[str, object]], device: int = 0) -> Engine:
    eng = Engine(cfg, device)
    eng.load_weights(items)
    return eng
