"""Checkpoint key layout of the reference `MeshAnything` module and a seeded synthetic checkpoint.

The layout is what `main.py:99-104` (`load_state_dict(strict=True)`) expects; shapes were obtained by
constructing the reference's own modules.  There is no network in the build
environment, hence no released checkpoint: parity and throughput runs use `synthetic_state_dict()`,
a deterministic random initialisation in exactly this key layout (numpy PCG64 keyed by tensor name,
so any subset of tensors can be regenerated independently and identically on any box with this image).

The initialisation is *not* HF's N(0, 0.02): it is chosen so that attention scores, LayerNorm affine
terms and biases are all non-trivial (a kernel that drops a bias, a gamma or a softmax scale fails
parity), see `_KIND_INIT`.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Iterator, Tuple

import numpy as np

from .config import MAConfig

Spec = "OrderedDict[str, Tuple[Tuple[int, ...], str]]"

PE = "point_encoder.model."
SM = PE + "shape_model."
DEC = "transformer.model.decoder."
TOK = "tokenizer."


def _miche_block(prefix: str, W: int, out: "OrderedDict") -> None:
    # transformer_blocks.py:77-115 (ResidualAttentionBlock), 18-45 (c_qkv no bias: qkv_bias=false), 229-244 (MLP)
    out[prefix + "attn.c_qkv.weight"] = ((3 * W, W), "w_attn")
    out[prefix + "attn.c_proj.weight"] = ((W, W), "w_half")
    out[prefix + "attn.c_proj.bias"] = ((W,), "bias")
    out[prefix + "ln_1.weight"] = ((W,), "ln_w")
    out[prefix + "ln_1.bias"] = ((W,), "ln_b")
    out[prefix + "mlp.c_fc.weight"] = ((4 * W, W), "w_half")
    out[prefix + "mlp.c_fc.bias"] = ((4 * W,), "bias")
    out[prefix + "mlp.c_proj.weight"] = ((W, 4 * W), "w_half")
    out[prefix + "mlp.c_proj.bias"] = ((W,), "bias")
    out[prefix + "ln_2.weight"] = ((W,), "ln_w")
    out[prefix + "ln_2.bias"] = ((W,), "ln_b")


def _miche_cross_block(prefix: str, W: int, out: "OrderedDict") -> None:
    # transformer_blocks.py:188-226 (ResidualCrossAttentionBlock), 118-152
    out[prefix + "attn.c_q.weight"] = ((W, W), "w_attn")
    out[prefix + "attn.c_kv.weight"] = ((2 * W, W), "w_attn")
    out[prefix + "attn.c_proj.weight"] = ((W, W), "w_half")
    out[prefix + "attn.c_proj.bias"] = ((W,), "bias")
    for i in (1, 2, 3):
        out[prefix + f"ln_{i}.weight"] = ((W,), "ln_w")
        out[prefix + f"ln_{i}.bias"] = ((W,), "ln_b")
    out[prefix + "mlp.c_fc.weight"] = ((4 * W, W), "w_half")
    out[prefix + "mlp.c_fc.bias"] = ((4 * W,), "bias")
    out[prefix + "mlp.c_proj.weight"] = ((W, 4 * W), "w_half")
    out[prefix + "mlp.c_proj.bias"] = ((W,), "bias")


def state_dict_spec(cfg: MAConfig, include_unused: bool = True, bert_fused: bool = False) -> "OrderedDict":
    """name -> (shape, init kind) for every tensor of the reference state dict.

    include_unused: also list tensors that exist in the reference checkpoint but are never read on the
        hot path (`shape_projection`, `geo_decoder.*`, `embed_tokens`, the logvar half of `pre_kl`).
    bert_fused: name the detokenizer layers the way optimum's BetterTransformer does (`in_proj_weight`, ...)
        instead of the vanilla HF BERT names.
    """
    W, T, H, E = cfg.enc_width, cfg.cond_length, cfg.hidden, cfg.embed_dim
    s: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    # ---- A.1 point_encoder ----
    if include_unused:
        s[PE + "shape_projection"] = ((W, W), "unused")           # clip_asl_module.py:24 (torch.empty)
    s[SM + "encoder.query"] = ((T, W), "table")                   # sal_perceiver.py:42
    s[SM + "encoder.input_proj.weight"] = ((W, cfg.point_in_dim), "w")
    s[SM + "encoder.input_proj.bias"] = ((W,), "bias")
    _miche_cross_block(SM + "encoder.cross_attn.", W, s)
    for n in range(cfg.enc_layers):
        _miche_block(SM + f"encoder.self_attn.resblocks.{n}.", W, s)
    s[SM + "encoder.ln_post.weight"] = ((W,), "ln_w")
    s[SM + "encoder.ln_post.bias"] = ((W,), "ln_b")
    s[SM + "pre_kl.weight"] = ((2 * E, W), "w")                   # rows [0,E) mean, [E,2E) logvar (unused by mode())
    s[SM + "pre_kl.bias"] = ((2 * E,), "bias")
    s[SM + "post_kl.weight"] = ((W, E), "w")
    s[SM + "post_kl.bias"] = ((W,), "bias")
    for n in range(cfg.shape_layers):
        _miche_block(SM + f"transformer.resblocks.{n}.", W, s)
    if include_unused:                                            # sal_perceiver.py:113-158, never run by MeshAnything.forward
        g = SM + "geo_decoder."
        s[g + "query_proj.weight"] = ((W, cfg.fourier_dim), "unused")
        s[g + "query_proj.bias"] = ((W,), "unused")
        tmp: "OrderedDict" = OrderedDict()
        _miche_cross_block(g + "cross_attn_decoder.", W, tmp)
        for k, (shp, _) in tmp.items():
            s[k] = (shp, "unused")
        s[g + "ln_post.weight"] = ((W,), "unused")
        s[g + "ln_post.bias"] = ((W,), "unused")
        s[g + "output_proj.weight"] = ((1, W), "unused")
        s[g + "output_proj.bias"] = ((1,), "unused")
    # ---- A.2 transformer (ShapeOPT) ----
    if include_unused:
        s[DEC + "embed_tokens.weight"] = ((cfg.vocab, H), "unused")   # shape_opt.py:199 "not used"
    s[DEC + "extra_embeds.weight"] = ((3, H), "table")
    s[DEC + "input_layer.weight"] = ((H, cfg.codebook_dim), "w")
    s[DEC + "input_layer.bias"] = ((H,), "bias")
    s[DEC + "embed_positions.weight"] = ((cfg.max_positions + 2, H), "pos")
    s[DEC + "token_embed_positions.weight"] = ((cfg.face_per_token + 3, H), "pos")
    s[DEC + "cond_embed.weight"] = ((2, H), "pos")
    for n in range(cfg.layers):
        p = DEC + f"layers.{n}."
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"self_attn.{proj}.weight"] = ((H, H), "w_attn" if proj != "out_proj" else "w_res")
            s[p + f"self_attn.{proj}.bias"] = ((H,), "bias")
        s[p + "self_attn_layer_norm.weight"] = ((H,), "ln_w")
        s[p + "self_attn_layer_norm.bias"] = ((H,), "ln_b")
        s[p + "fc1.weight"] = ((cfg.ffn, H), "w")
        s[p + "fc1.bias"] = ((cfg.ffn,), "bias")
        s[p + "fc2.weight"] = ((H, cfg.ffn), "w_res")
        s[p + "fc2.bias"] = ((H,), "bias")
        s[p + "final_layer_norm.weight"] = ((H,), "ln_w")
        s[p + "final_layer_norm.bias"] = ((H,), "ln_b")
    s[DEC + "quantize_codebooks"] = ((1, cfg.codebook_size, cfg.codebook_dim), "codebook")  # meshanything.py:118
    s["transformer.lm_head.weight"] = ((cfg.vocab, H), "w_head")  # shape_opt.py:24, un-tied (29-43)
    # ---- A.4 top level ----
    s["cond_head_proj.weight"] = ((H, W), "w")
    s["cond_head_proj.bias"] = ((H,), "bias")
    s["cond_proj.weight"] = ((H, 2 * W), "w")
    s["cond_proj.bias"] = ((H,), "bias")
    # ---- A.3 tokenizer (NoiseResistantDecoder) ----
    Wt = cfg.tok_width
    s[TOK + "pos_embedding.weight"] = ((cfg.tok_max_pos, Wt), "pos")
    s[TOK + "point_pe.weight"] = ((T, Wt), "pos")
    s[TOK + "layernorm.weight"] = ((Wt,), "ln_w")
    s[TOK + "layernorm.bias"] = ((Wt,), "ln_b")
    s[TOK + "point_layernorm.weight"] = ((Wt,), "ln_w")
    s[TOK + "point_layernorm.bias"] = ((Wt,), "ln_b")
    s[TOK + "cond_proj.weight"] = ((Wt, W), "w")
    s[TOK + "cond_proj.bias"] = ((Wt,), "bias")
    s[TOK + "cond_head_proj.weight"] = ((Wt, W), "w")
    s[TOK + "cond_head_proj.bias"] = ((Wt,), "bias")
    s[TOK + "project_down_codebook.weight"] = ((Wt, 3 * cfg.codebook_dim), "w")
    s[TOK + "project_down_codebook.bias"] = ((Wt,), "bias")
    s[TOK + "to_coor_logits.0.weight"] = ((9 * cfg.discrete_num, Wt), "w_head")
    s[TOK + "to_coor_logits.0.bias"] = ((9 * cfg.discrete_num,), "bias")
    for n in range(cfg.tok_layers):
        p = TOK + f"decoder.layer.{n}."
        if bert_fused:   # optimum BertLayerBetterTransformer parameter names
            s[p + "in_proj_weight"] = ((3 * Wt, Wt), "w_attn")
            s[p + "in_proj_bias"] = ((3 * Wt,), "bias")
            s[p + "out_proj_weight"] = ((Wt, Wt), "w")
            s[p + "out_proj_bias"] = ((Wt,), "bias")
            s[p + "linear1_weight"] = ((cfg.tok_ffn, Wt), "w")
            s[p + "linear1_bias"] = ((cfg.tok_ffn,), "bias")
            s[p + "linear2_weight"] = ((Wt, cfg.tok_ffn), "w")
            s[p + "linear2_bias"] = ((Wt,), "bias")
            s[p + "norm1_weight"] = ((Wt,), "ln_w")
            s[p + "norm1_bias"] = ((Wt,), "ln_b")
            s[p + "norm2_weight"] = ((Wt,), "ln_w")
            s[p + "norm2_bias"] = ((Wt,), "ln_b")
        else:            # vanilla HF BertLayer names
            for nm in ("query", "key", "value"):
                s[p + f"attention.self.{nm}.weight"] = ((Wt, Wt), "w_attn")
                s[p + f"attention.self.{nm}.bias"] = ((Wt,), "bias")
            s[p + "attention.output.dense.weight"] = ((Wt, Wt), "w")
            s[p + "attention.output.dense.bias"] = ((Wt,), "bias")
            s[p + "attention.output.LayerNorm.weight"] = ((Wt,), "ln_w")
            s[p + "attention.output.LayerNorm.bias"] = ((Wt,), "ln_b")
            s[p + "intermediate.dense.weight"] = ((cfg.tok_ffn, Wt), "w")
            s[p + "intermediate.dense.bias"] = ((cfg.tok_ffn,), "bias")
            s[p + "output.dense.weight"] = ((Wt, cfg.tok_ffn), "w")
            s[p + "output.dense.bias"] = ((Wt,), "bias")
            s[p + "output.LayerNorm.weight"] = ((Wt,), "ln_w")
            s[p + "output.LayerNorm.bias"] = ((Wt,), "ln_b")
    return s


# kind -> (distribution, parameters).  "fan" = std gain/sqrt(fan_in).
_KIND_INIT = {
    "w": ("fan", 1.0),
    "w_attn": ("fan", 1.0),     # q/k/v projections: unit-variance q,k -> O(1) attention scores
    "w_half": ("fan", 0.5),     # miche proj / MLP: keeps the pre-LN residual stream O(1) over 24 blocks
    "w_head": ("fan", 1.0),     # lm_head / to_coor_logits: O(1) logits
    "w_res": ("fan", 1.0),      # the decoder's residual-branch outputs (out_proj, fc2): see init="diverse"
    "bias": ("normal", 0.05),
    "ln_w": ("affine", (1.0, 0.1)),
    "ln_b": ("normal", 0.05),
    "table": ("normal", 0.5),
    "pos": ("normal", 0.3),
    "codebook": ("normal", 1.0),
    "unused": ("normal", 0.01),
}


# init="hf": what the reference's own constructors would leave before training: transformers' `_init_weights`
# for the OPT decoder / BERT detokenizer / top-level projections (Linear and Embedding N(0, 0.02), biases 0, LayerNorm (1, 0)),
# Michelangelo's `init_linear` (std 0.25 / sqrt(width), transformer_blocks.py:12-15) for the point encoder, codebook N(0, 1)
# (meshanything.py:118 loads a trained codebook; unit variance is its scale).  Logit margins are ~5x smaller than under the
# default init: used by the report-only fidelity tests, not by parity gates.
def _hf_init(cfg: MAConfig, name: str, kind: str):
    miche = name.startswith(PE)
    if kind in ("w", "w_attn", "w_half", "w_head", "w_res"):
        return ("normal", 0.25 / float(np.sqrt(cfg.enc_width)) if miche else 0.02)
    if kind == "bias" or kind == "ln_b":
        return ("normal", 0.0)
    if kind == "ln_w":
        return ("affine", (1.0, 0.0))
    if kind in ("table", "pos"):
        return ("normal", 0.02)
    return _KIND_INIT[kind]


# init="diverse": the default init with the decoder's residual branches (out_proj, fc2) at a fifth of their gain -- the SAME random
# numbers, scaled.  Why: a 24-layer post-LN ReLU transformer at a unit-gain random init maps every input to nearly the same output (each
# ReLU MLP adds a constant mean vector, each attention an average over the 257 prefix rows: the correlation between different inputs
# tends to 1 with depth), so greedy decoding sits in a fixed point -- the default checkpoint emits token 2668 for ever, the HF-style one
# cycles through 3 ids -- and a test on such a stream sees one embedding row, one slot pattern, one argmax.  With gain 0.2 every layer
# still moves the logits by far more than any tolerance used here, but the stream depends on its own tokens and positions: ~50 distinct
# ids in 160 greedy steps at the 350M shape (fp32 top-1/top-2 margin: median 0.15, 10 % quantile 0.02).  The 350M greedy tests, the
# greedy reference anchor (tests/golden/full_anchor_hf.npz) and bench.py use it.
_DIVERSE_RES_GAIN = 0.2


def _tensor_rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(name.encode())]))


def synthetic_tensor(cfg: MAConfig, name: str, shape: Tuple[int, ...], kind: str, seed: int = 1234, init: str = "default") -> np.ndarray:
    rng = _tensor_rng(seed, name)
    dist, par = _hf_init(cfg, name, kind) if init == "hf" else _KIND_INIT[kind]
    if init == "diverse" and kind == "w_res":
        par = par * _DIVERSE_RES_GAIN
    x = rng.standard_normal(shape, dtype=np.float32)
    if dist == "fan":
        fan_in = shape[-1]
        x *= np.float32(par / np.sqrt(fan_in))
    elif dist == "normal":
        x *= np.float32(par)
    elif dist == "affine":
        x = np.float32(par[0]) + np.float32(par[1]) * x
    return np.ascontiguousarray(x, dtype=np.float32)


def synthetic_items(cfg: MAConfig, seed: int = 1234, include_unused: bool = False,
                    bert_fused: bool = False, init: str = "default") -> Iterator[Tuple[str, np.ndarray]]:
    """Yield (reference key, fp32 ndarray) one tensor at a time (the 350M layout is 2.4 GB in fp32)."""
    assert init in ("default", "hf", "diverse")
    for name, (shape, kind) in state_dict_spec(cfg, include_unused, bert_fused).items():
        yield name, synthetic_tensor(cfg, name, shape, kind, seed, init)


def synthetic_state_dict(cfg: MAConfig, seed: int = 1234, include_unused: bool = False,
                         bert_fused: bool = False, init: str = "default") -> "OrderedDict[str, np.ndarray]":
    return OrderedDict(synthetic_items(cfg, seed, include_unused, bert_fused, init))


def load_safetensors_items(path: str) -> Iterator[Tuple[str, object]]:
    """Iterate the tensors of the released checkpoint (`MeshAnything_350m.pth` is a safetensors file, main.py:95-104).
    Tensors come back as torch tensors in their STORED dtype (numpy has no bfloat16, so the numpy reader rejects bf16 files);
    the engine's packer takes fp32 / fp16 / bf16 storage as is (Engine._desc)."""
    from safetensors import safe_open
    with safe_open(path, framework="pt", device="cpu") as f:
        for k in f.keys():
            yield k, f.get_tensor(k)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (the engine's bf16 policy), NaN-preserving."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32).copy()
    nan = np.isnan(x)
    if nan.any():
        out[nan] = x[nan]
    return out.reshape(x.shape)
