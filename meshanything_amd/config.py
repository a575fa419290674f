"""Static shape description of the MeshAnything hot path.

The reference hard-codes these numbers in three places: the Michelangelo YAML
(`MeshAnything/miche/shapevae-256.yaml:7-19`), the HF configs it downloads
(`facebook/opt-350m`, `bert-base-uncased`; `MeshAnything/models/meshanything.py:22-23,95-113`)
and literals in `meshanything.py:27-41,88-98`.  The engine takes them as run-time
dimensions so that the same kernels run a tiny configuration (oracle finishes in
seconds) and the 350M configuration (BASELINE.json).  `head_dim` is 64 everywhere
(768/12, 1024/16) and the kernels rely on that.

`MAConfig.to_c()` produces the `ma_config` struct of `include/meshanything_amd.h`.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, asdict, fields

HEAD_DIM = 64

# weight / KV-cache precision policies (include/meshanything_amd.h: MA_DTYPE_*)
DTYPE_F32 = 0   # "exact" mode: fp32 weights, fp32 KV cache, no activation rounding
DTYPE_BF16 = 1  # bf16 weights + bf16 KV cache; GEMM/attention inputs rounded to bf16, fp32 accumulate
DTYPE_F16 = 2   # the same with IEEE half: the reference's own arithmetic (fp16 autocast, main.py:114-118,149)


@dataclass
class MAConfig:
    # ---- point encoder (Michelangelo perceiver; shapevae-256.yaml:7-19) ----
    n_points: int = 4096        # main.py:25 (Dataset samples exactly 4096 points)
    num_freqs: int = 8          # yaml num_freqs (include_pi: false)
    enc_width: int = 768        # yaml width
    enc_heads: int = 12         # yaml heads
    num_latents: int = 256      # yaml num_latents (+1 shape token, sal_perceiver.py:332)
    enc_layers: int = 8         # yaml num_encoder_layers (self-attn blocks after the cross block)
    shape_layers: int = 16      # yaml num_decoder_layers (`transformer`, used by to_shape_latents)
    embed_dim: int = 64         # yaml embed_dim (pre_kl -> 2*embed_dim, mode() keeps the first half)
    # ---- autoregressive decoder (ShapeOPT, facebook/opt-350m shape) ----
    hidden: int = 1024
    heads: int = 16
    layers: int = 24
    ffn: int = 4096
    codebook_size: int = 8192   # main.py:77
    codebook_dim: int = 1024    # main.py:78
    n_max_faces: int = 800      # main.py:80 (--n_max_triangles)
    max_positions: int = 18259  # meshanything.py:97-98 (embed_positions has +2 offset rows)
    # ---- detokenizer (NoiseResistantDecoder, bert-base-uncased shape, 6 layers) ----
    tok_width: int = 768
    tok_heads: int = 12
    tok_layers: int = 6
    tok_ffn: int = 3072
    tok_max_pos: int = 18000    # meshanything.py:27
    discrete_num: int = 128     # meshanything.py:18
    # ---- engine policy ----
    max_batch: int = 1
    dtype: int = DTYPE_BF16
    kv_splits: int = 0          # reserved (the decode attention always splits a head's cache into 16 equal chunks)
    use_graph: int = 1          # capture one decode step in a hipGraph and replay it
    enc_exact: int = 1          # 16-bit policies: the point encoder (encode_latents + process_point_feature) stays fp32 (1e-5 on its activations)

    # ---- derived ----
    @property
    def cond_length(self) -> int:          # meshanything.py:90 (257)
        return self.num_latents + 1

    @property
    def vocab(self) -> int:                # meshanything.py:99 (codebook + bos/eos/pad)
        return self.codebook_size + 3

    @property
    def face_per_token(self) -> int:       # meshanything.py:88-89 (3 quantizers x 3 vertices)
        return 9

    @property
    def max_new_tokens(self) -> int:       # meshanything.py:93,140 (n_max_faces*9 + 2)
        return self.n_max_faces * 9 + 2

    @property
    def max_seq(self) -> int:              # prefix + generated
        return self.cond_length + self.max_new_tokens

    @property
    def fourier_dim(self) -> int:          # embedder.py:78-85: 3*(2*num_freqs+1)
        return 3 * (2 * self.num_freqs + 1)

    @property
    def point_in_dim(self) -> int:         # fourier + 3 normals (sal_perceiver.py:45)
        return self.fourier_dim + 3

    def validate(self) -> None:
        assert self.enc_width == self.enc_heads * HEAD_DIM
        assert self.hidden == self.heads * HEAD_DIM
        assert self.tok_width == self.tok_heads * HEAD_DIM
        assert self.codebook_dim == self.hidden, "word_embed_proj_dim is forced to hidden (meshanything.py:112-113)"
        assert self.max_seq <= self.max_positions
        assert self.n_max_faces <= self.tok_max_pos
        for d in (self.enc_width, self.hidden, self.ffn, self.tok_width, self.tok_ffn, self.embed_dim, self.codebook_dim):
            assert d % 32 == 0, f"GEMM K/N dims must be multiples of 32, got {d}"
        assert self.point_in_dim <= 64

    @staticmethod
    def full(**kw) -> "MAConfig":
        """The 350M checkpoint's shape (BASELINE.json configs 1-4)."""
        return MAConfig(**kw)

    @staticmethod
    def tiny(**kw) -> "MAConfig":
        """A small configuration with the same structure (parity tests; the oracle runs it in seconds)."""
        base = dict(n_points=256, num_freqs=8, enc_width=128, enc_heads=2, num_latents=16, enc_layers=2,
                    shape_layers=2, embed_dim=32, hidden=128, heads=2, layers=2, ffn=256,
                    codebook_size=61, codebook_dim=128, n_max_faces=8, max_positions=17 + 8 * 9 + 2 + 6,
                    tok_width=128, tok_heads=2, tok_layers=2, tok_ffn=256, tok_max_pos=32, discrete_num=128)
        base.update(kw)
        return MAConfig(**base)

    def to_c(self) -> "CMAConfig":
        self.validate()
        c = CMAConfig()
        c.struct_size = ctypes.sizeof(CMAConfig)
        for f in fields(self):
            setattr(c, f.name, int(getattr(self, f.name)))
        return c

    def as_dict(self) -> dict:
        return asdict(self)


class CMAConfig(ctypes.Structure):
    """Mirror of `struct ma_config` (include/meshanything_amd.h).  Field order must match."""
    _fields_ = [("struct_size", ctypes.c_int32)] + [(f.name, ctypes.c_int32) for f in fields(MAConfig)]
