"""The reference's Python call surface on top of the HIP engine.

A user of buaacyw/MeshAnything constructs `MeshAnything(args)`, calls `load_state_dict(tensors, strict=True)` and then
`model(pc_normal, sampling=...)` (main.py:91-104,152).  This module keeps those names and meanings; every method is a
thin call into libmeshanything_amd.so through `Engine` (no arithmetic happens in Python, and there is no fallback).

    reference call                                            (file:line)                      here
    MeshAnything(args)                                        meshanything.py:83-123           MeshAnything(args)
    model.load_state_dict(tensors, strict=True)               main.py:99-104                   MeshAnything.load_state_dict
    model(pc_normal, sampling)                                meshanything.py:134-176          MeshAnything.forward
    model.point_encoder.encode_latents(pc)                    asl_pl_module.py:145-157         PointEncoder.encode_latents
    model.point_encoder.to_shape_latents(latents)             asl_pl_module.py:182-185         PointEncoder.to_shape_latents
    model.process_point_feature(point_feature)                meshanything.py:125-132          MeshAnything.process_point_feature
    model.transformer.generate(inputs_embeds=..., ...)        meshanything.py:143-162          Transformer.generate
    model.get_codes(indices)                                  meshanything.py:178-212          MeshAnything.get_codes
    model.tokenizer(ids, codes, point_feature=...)            meshanything.py:50-80            Tokenizer.__call__

`args` needs the attributes the reference reads: `.llm` (config name only; ignored like the reference ignores its
weights), `.codebook_size`, `.codebook_dim`, `.n_max_triangles` (meshanything.py:19-20,93,96).  Extra, optional:
`.dtype` ("bf16" | "fp16" | "fp32"), `.batchsize_per_gpu` (engine max_batch), `.device` (GPU index).
"""
from __future__ import annotations

import os
from typing import Dict, Mapping, Optional

import torch

from .config import DTYPE_BF16, DTYPE_F16, DTYPE_F32, MAConfig
from .engine import Engine

BOS_TOKEN_ID, EOS_TOKEN_ID, PAD_TOKEN_ID = 0, 1, 2        # meshanything.py:102-104


def config_from_args(args) -> MAConfig:
    if isinstance(getattr(args, "ma_config", None), MAConfig):      # tests / tooling: an explicit shape (e.g. MAConfig.tiny())
        return args.ma_config
    dtype = {"bf16": DTYPE_BF16, "fp16": DTYPE_F16, "fp32": DTYPE_F32}[getattr(args, "dtype", "bf16")]
    return MAConfig.full(codebook_size=int(getattr(args, "codebook_size", 8192)), codebook_dim=int(getattr(args, "codebook_dim", 1024)),
                         n_max_faces=int(getattr(args, "n_max_triangles", 800)), max_batch=int(getattr(args, "batchsize_per_gpu", 1)),
                         dtype=dtype)


class PointEncoder:
    """`model.point_encoder` (AlignedShapeAsLatentPLModule): the two methods the hot path calls."""

    def __init__(self, engine: Engine):
        self._e = engine

    def encode_latents(self, surface: torch.Tensor) -> torch.Tensor:
        """(B, N, 6) xyz+normal -> (B, 257, 768): cat(shape_embed, latents) (asl_pl_module.py:145-157)."""
        latents, _ = self._e.encode(surface, want_prefix=False)
        return latents

    def to_shape_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """(B, 256, 768) -> (B, 256, 768) (asl_pl_module.py:182-185)."""
        assert latents.shape[1] == self._e.cfg.num_latents, "to_shape_latents expects the 256 latent tokens (asl_pl_module.py:154)"
        return self._e.to_shape_latents(latents)


class Transformer:
    """`model.transformer` (ShapeOPT): `generate` with the arguments the reference passes (meshanything.py:143-162)."""

    def __init__(self, engine: Engine):
        self._e = engine

    def generate(self, inputs_embeds: torch.Tensor, max_new_tokens: Optional[int] = None, num_beams: int = 1, do_sample: bool = False,
                 top_k: int = 50, top_p: float = 0.95, bos_token_id: int = BOS_TOKEN_ID, eos_token_id: int = EOS_TOKEN_ID,
                 pad_token_id: int = PAD_TOKEN_ID, uniforms: Optional[torch.Tensor] = None, seed: int = 0) -> torch.Tensor:
        """-> LongTensor (B, n <= max_new_tokens) of NEW tokens only; finished rows padded with pad_token_id; generation
        stops when every row has emitted eos ([3p] GenerationMixin greedy / sample semantics)."""
        if num_beams != 1:
            raise NotImplementedError("the reference only calls generate() with num_beams=1 (meshanything.py:147)")
        if (bos_token_id, eos_token_id, pad_token_id) != (BOS_TOKEN_ID, EOS_TOKEN_ID, PAD_TOKEN_ID):
            raise ValueError("special token ids are fixed by the checkpoint: bos 0, eos 1, pad 2 (meshanything.py:102-104)")
        tokens, _ = self._e.generate(inputs_embeds, sampling=bool(do_sample), max_new_tokens=max_new_tokens, top_k=top_k, top_p=top_p,
                                     uniforms=uniforms, seed=seed)
        return tokens


class Tokenizer:
    """`model.tokenizer` (NoiseResistantDecoder, meshanything.py:10-80)."""

    pad_id = -1                                              # meshanything.py:15

    def __init__(self, engine: Engine, args):
        self._e = engine
        self.codebook_size = engine.cfg.codebook_size
        self.codebook_dim = engine.cfg.codebook_dim
        self.num_quantizers = 3

    def __call__(self, input_ids: torch.Tensor, input_embeds: Optional[torch.Tensor] = None, point_feature: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ids (B, 9F) in [-1, codebook), codes (B, 3F, D) (what get_codes returns, or any other embeddings of that shape),
        point_feature (B, 257, 768) -> (B, F, 3, 3) fp32 vertex coordinates, NaN rows = invalid faces.  `input_embeds` is used
        as the face codes exactly as the reference uses it (meshanything.py:53-55); None: get_codes(input_ids)."""
        if point_feature is None:
            raise ValueError("tokenizer(...) needs point_feature (the raw 257x768 encoder latents, meshanything.py:174)")
        return self._e.detokenize(input_ids, point_feature, codes=input_embeds)

    forward = __call__


class MeshAnything(torch.nn.Module):
    """Drop-in for `MeshAnything.models.meshanything.MeshAnything` (inference only, like the reference: eval + no_grad)."""

    def __init__(self, args, device: Optional[int] = None):
        super().__init__()
        self.args = args
        self.cfg = config_from_args(args)
        dev = device if device is not None else int(getattr(args, "device", torch.cuda.current_device() if torch.cuda.is_available() else 0))
        self.engine = Engine(self.cfg, dev)                  # raises without a GPU / without the HIP library
        self.point_encoder = PointEncoder(self.engine)
        self.transformer = Transformer(self.engine)
        self.tokenizer = Tokenizer(self.engine, args)
        self.num_quantizers = 3
        self.face_per_token = self.num_quantizers * 3
        self.cond_length = self.cfg.cond_length
        self.cond_dim = self.cfg.enc_width
        self.max_length = self.cfg.n_max_faces * self.face_per_token + 2 + self.cond_length
        self.bos_token_id, self.eos_token_id, self.pad_token_id = BOS_TOKEN_ID, EOS_TOKEN_ID, PAD_TOKEN_ID
        # sampling randomness: the reference seeds torch once (`set_seed(args.seed)`, main.py:129-133) and its CUDA RNG then
        # advances from batch to batch and differs per rank.  Here the sampler draws from a counter-hashed uniform stream
        # keyed by (seed, row, step), so every forward() call gets its own stream seed derived from (args.seed, rank, call #).
        self._seed = int(getattr(args, "seed", 0))
        self._rank = int(os.environ.get("RANK", "0"))
        self._calls = 0
        self.eval()

    # ---- weights -----------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Mapping[str, object], strict: bool = True):   # noqa: D401 (nn.Module signature)
        """main.py:99-104.  Keys are the reference's state-dict keys; tensors may live on any device.  strict=True
        (the reference's setting): every tensor the hot path needs must be present, unknown keys are an error."""
        unexpected = []
        if strict:
            self.engine.load_weights(state_dict.items(), finalize=True)
        else:
            from ._lib import MAError
            for k, v in state_dict.items():
                try:
                    self.engine.load_weights([(k, v)], finalize=False)
                except MAError as err:
                    if err.code != -4:                       # MA_ERR_UNKNOWN_TENSOR: skipped when not strict
                        raise
                    unexpected.append(k)
            self.engine.load_weights([], finalize=True)      # the hot path still needs every tensor it reads
        return torch.nn.modules.module._IncompatibleKeys([], unexpected)

    # ---- the hot path ------------------------------------------------------------------------------------------
    def process_point_feature(self, point_feature: torch.Tensor) -> torch.Tensor:
        return self.engine.process_point_feature(point_feature)

    def get_codes(self, indices: torch.Tensor) -> torch.Tensor:
        return self.engine.get_codes(indices)

    def _next_stream_seed(self) -> int:
        """splitmix64 of (args.seed, rank, number of forward() calls so far): reproducible, distinct per call and per rank."""
        z = (self._seed * 0x9E3779B97F4A7C15 + self._rank * 0xD1B54A32D192ED03 + self._calls * 0x94D049BB133111EB + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        self._calls += 1
        return z ^ (z >> 31)

    @torch.no_grad()
    def forward(self, pc_normal: torch.Tensor, sampling: bool = False, seed: Optional[int] = None) -> torch.Tensor:
        """(B, 4096, 6) -> (B, n_max_triangles, 3, 3) fp32, NaN rows = invalid faces (meshanything.py:134-176): one library call.
        `seed` (optional) pins the sampler's uniform stream for this call; by default it advances from call to call."""
        return self.forward_detailed(pc_normal, sampling, seed=seed)["coords"]

    @torch.no_grad()
    def forward_detailed(self, pc_normal: torch.Tensor, sampling: bool = False, seed: Optional[int] = None, **kw) -> Dict[str, object]:
        """forward() plus the intermediate tensors (tokens, lengths, ids, latents) for tests and tooling."""
        s = self._next_stream_seed() if seed is None else int(seed)
        return self.engine.forward(pc_normal, sampling=bool(sampling), seed=s, **kw)
