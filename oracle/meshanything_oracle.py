"""Oracle for the MeshAnything hot path (plain PyTorch, CPU by default) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module.  The product (`meshanything_amd/`) never does; it fails loudly without its HIP library.

What this is: a plain PyTorch-CPU fp32 restatement of `MeshAnything.forward`
(`/root/reference/MeshAnything/models/meshanything.py:134-176`) and everything it reaches, written
against the reference's state-dict key layout (`meshanything_amd/checkpoint.py`).  Every function
cites the reference lines it follows.  Third-party arithmetic that is not under `/root/reference`
(transformers==4.39.3 `OPTDecoderLayer` / `OptFlashAttention2` / `OPTLearnedPositionalEmbedding` /
`GenerationMixin.generate` and its logits warpers, BERT layers) is restated from the published
algorithm; where the container holds a newer copy the restatement cites it.

PIN STATUS: every function the reference's own code can execute here is pinned against that code's output (below); the
`generate()` loop semantics are pinned against the container's HuggingFace `GenerationMixin.generate` driving this
oracle's step function (tests/test_generate_loop_vs_hf.py); the reference's own ShapeOPTDecoder.forward is pinned through
tests/golden/shapeopt_forward.npz (prefix call + cached steps); the OUTERMOST composition -- the reference's ShapeOPT CausalLM
wrapper under the container's GenerationMixin.generate with the call of meshanything.py:143-151 -- through
tests/golden/shapeopt_generate.npz (tests/test_reference_anchor.py; three documented version shims, none in the arithmetic); the
350M-shape logits and detokenizer bins through tests/golden/full_anchor.npz.  Outside the pins: transformers==4.39.3 itself and
flash-attn (absent from this container; eager attention behind a 4.39.3-signature adapter).

How it is pinned: the reference has no tests and no golden vectors, so the oracle
is pinned against outputs of the reference's *own code* run in the authoring container
(`tests/golden/make_golden.py` imports `/root/reference` and writes `tests/golden/*.npz`;
`tests/test_oracle_golden.py` compares).  Pinned that way: Dataset normalisation, Fourier embedder,
`encode_latents`, `to_shape_latents`, `process_point_feature`, `embed_with_vae`,
`OPTFacePositionalEmbedding`, `get_codes`, `NoiseResistantDecoder.forward`, `undiscretize`;
and against the container's transformers copy: `OPTLearnedPositionalEmbedding`, `OPTDecoderLayer`
(post-LN) stack with KV cache, BERT layer, TopK/TopP warpers; `generate` as a whole against the reference's ShapeOPT wrapper
under GenerationMixin.generate (above).

Device (`device`): "cpu" (default; what the CPU suite, the golden pins and bench.py's cpu_baseline use) or a torch-ROCm device.
  On "cuda" the SAME statements below run as stock PyTorch fp32 ops (rocBLAS GEMMs with TF32-style shortcuts disabled, eager
  softmax / LayerNorm): still an implementation independent of the HIP engine, but the long verifications of the GPU suite
  (teacher-forced passes over thousands of tokens at the 350M shape) no longer depend on the GPU box's host cores -- the
  driver's un-tasksetted run of round 2 spent its whole 1200 s there.  Public methods take tensors from any device and return
  CPU tensors; tests/test_gpu_oracle_device.py cross-checks the two devices against each other.

Precision policy (`policy`):
  "fp32": no rounding anywhere -- the reference's CPU-equivalent arithmetic.
  "fp16": the same rounding points with IEEE half (the engine's MA_DTYPE_F16 mode: the reference's own fp16-autocast arithmetic class).
  "bf16": mirrors the engine's MA_DTYPE_BF16 mode so that comparisons are like-for-like:
          every Linear computes fp32-accumulated dot products of bf16(x) and bf16(W);
          attention uses bf16(q), bf16(k), bf16(v) with fp32 scores/softmax/accumulation;
          everything else (bias, LayerNorm, GELU/ReLU, residuals, embedding tables) is fp32.
          With cfg.enc_exact (the default) the point encoder -- encode_latents, to_shape_latents, process_point_feature and the
          detokenizer's projection of the latents -- is NOT rounded: the engine keeps it in fp32 under a 16-bit policy.
"""
from __future__ import annotations

import functools
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from meshanything_amd.config import MAConfig
from meshanything_amd.checkpoint import PE, SM, DEC, TOK

BOS, EOS, PAD = 0, 1, 2          # meshanything.py:102-104
PAD_ID = -1                      # NoiseResistantDecoder.pad_id, meshanything.py:15


def bf16r(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 (round-to-nearest-even) -> fp32."""
    return x.to(torch.bfloat16).to(torch.float32)


def fp16r(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> IEEE half (round-to-nearest-even, overflow to inf) -> fp32."""
    return x.to(torch.float16).to(torch.float32)


ROUND16 = {"bf16": bf16r, "fp16": fp16r}


def undiscretize(t: torch.Tensor, low: float, high: float, num_discrete: int) -> torch.Tensor:
    """meshanything.py:214-223."""
    t = t.float()
    t = t / num_discrete
    return t * (high - low) + low


def normalize_pc(pc_normal: np.ndarray) -> np.ndarray:
    """Dataset.__getitem__, main.py:45-58: centre on the bbox mid-point, scale to +-0.9995, cast fp16.

    Arithmetic is numpy in the input's dtype (fp16 stays fp16 for an fp16 .npy), as in the reference."""
    pc_coor = pc_normal[:, :3]
    normals = pc_normal[:, 3:]
    bounds = np.array([pc_coor.min(axis=0), pc_coor.max(axis=0)])
    pc_coor = pc_coor - (bounds[0] + bounds[1])[None, :] / 2
    pc_coor = pc_coor / np.abs(pc_coor).max() * 0.9995
    assert (np.linalg.norm(normals, axis=-1) > 0.99).all(), "normals should be unit vectors, something wrong"
    return np.concatenate([pc_coor, normals], axis=-1, dtype=np.float16)


def sample_points(cur_data: np.ndarray, n: int = 4096) -> np.ndarray:
    """Dataset.__init__, main.py:23-26: np.random.choice without replacement from the global numpy RNG."""
    assert cur_data.shape[0] >= n, "input pc_normal should have at least 4096 points"
    idx = np.random.choice(cur_data.shape[0], n, replace=False)
    return cur_data[idx]


def _move(x, device):
    """Tensors (also inside tuples / lists / dicts) -> device; everything else untouched."""
    if isinstance(x, torch.Tensor):
        return x.to(device)
    if isinstance(x, tuple):
        return tuple(_move(v, device) for v in x)
    if isinstance(x, list):
        return [_move(v, device) for v in x]
    if isinstance(x, dict):
        return {k: _move(v, device) for k, v in x.items()}
    return x


def _api(fn):
    """Public entry point of the oracle: tensor arguments go to the oracle's device, the OUTERMOST call hands its results back
    on the CPU (calls between oracle methods stay on the device)."""
    @functools.wraps(fn)
    def wrap(self, *args, **kw):
        if self.device.type == "cpu":
            return fn(self, *args, **kw)
        if self._depth == 0:            # containers (a KV-cache list updated in place) keep their identity
            args = [a.to(self.device) if isinstance(a, torch.Tensor) else a for a in args]
            kw = {k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
        self._depth += 1
        try:
            r = fn(self, *args, **kw)
        finally:
            self._depth -= 1
        return _move(r, "cpu") if (self._depth == 0 and self._keep == 0) else r
    return wrap


class Oracle:
    def __init__(self, cfg: MAConfig, state_dict: Dict[str, np.ndarray], policy: str = "fp32", device: str = "cpu"):
        assert policy in ("fp32", "bf16", "fp16")
        self.cfg = cfg
        self.policy = policy
        self.device = torch.device(device)
        self._depth = 0                           # nesting of public calls (arguments are moved by the outermost one only)
        self._keep = 0                            # > 0 inside `with oracle.on_device()`: results stay on the oracle's device
        if self.device.type == "cuda":            # plain fp32 arithmetic in the library GEMMs the torch ops reach
            torch.backends.cuda.matmul.allow_tf32 = False
            torch.backends.cudnn.allow_tf32 = False
            torch.set_float32_matmul_precision("highest")
        self.sd: Dict[str, torch.Tensor] = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)).to(self.device)
                                            for k, v in state_dict.items()}
        self._wcache: Dict[str, torch.Tensor] = {}
        self._accept_bert_fused()

    def on_device(self):
        """Context manager: oracle methods called inside keep their results on the oracle's device (the verifiers below reduce
        thousands of logit rows there instead of on the host)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self._keep += 1
            try:
                yield self
            finally:
                self._keep -= 1
        return cm()

    # ------------------------------------------------------------------ primitives
    def _accept_bert_fused(self) -> None:
        """Accept optimum BetterTransformer names for the detokenizer layers by un-fusing them."""
        for n in range(self.cfg.tok_layers):
            p = TOK + f"decoder.layer.{n}."
            if p + "in_proj_weight" in self.sd:
                Wt = self.cfg.tok_width
                w, b = self.sd[p + "in_proj_weight"], self.sd[p + "in_proj_bias"]
                for i, nm in enumerate(("query", "key", "value")):
                    self.sd[p + f"attention.self.{nm}.weight"] = w[i * Wt:(i + 1) * Wt].contiguous()
                    self.sd[p + f"attention.self.{nm}.bias"] = b[i * Wt:(i + 1) * Wt].contiguous()
                ren = {"out_proj_weight": "attention.output.dense.weight", "out_proj_bias": "attention.output.dense.bias",
                       "linear1_weight": "intermediate.dense.weight", "linear1_bias": "intermediate.dense.bias",
                       "linear2_weight": "output.dense.weight", "linear2_bias": "output.dense.bias",
                       "norm1_weight": "attention.output.LayerNorm.weight", "norm1_bias": "attention.output.LayerNorm.bias",
                       "norm2_weight": "output.LayerNorm.weight", "norm2_bias": "output.LayerNorm.bias"}
                for a, b2 in ren.items():
                    self.sd[p + b2] = self.sd[p + a]

    def _enc(self):
        """Context: the precision of the point encoder (fp32 under a 16-bit policy when cfg.enc_exact is set)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            saved = self.policy
            if getattr(self.cfg, "enc_exact", 0):
                self.policy = "fp32"
            try:
                yield self
            finally:
                self.policy = saved
        return cm()

    def W(self, name: str, rows: Optional[slice] = None) -> torch.Tensor:
        key = (name if rows is None else f"{name}[{rows.start}:{rows.stop}]") + "|" + self.policy
        w = self._wcache.get(key)
        if w is None:
            w = self.sd[name]
            if rows is not None:
                w = w[rows]
            if self.policy in ROUND16:
                w = ROUND16[self.policy](w)
            self._wcache[key] = w = w.contiguous()
        return w

    def rin(self, x: torch.Tensor) -> torch.Tensor:
        """Rounding applied to a GEMM / attention input under the active policy."""
        return ROUND16[self.policy](x) if self.policy in ROUND16 else x

    def linear(self, x: torch.Tensor, wname: str, bname: Optional[str] = None, rows: Optional[slice] = None) -> torch.Tensor:
        y = self.rin(x) @ self.W(wname, rows).t()
        if bname is not None:
            b = self.sd[bname]
            y = y + (b[rows] if rows is not None else b)
        return y

    def ln(self, x: torch.Tensor, prefix: str, eps: float, wkey: str = "weight", bkey: str = "bias") -> torch.Tensor:
        return F.layer_norm(x, (x.shape[-1],), self.sd[prefix + wkey], self.sd[prefix + bkey], eps)

    def attention(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float,
                  causal_offset: Optional[int] = None, q_chunk: int = 512, dense: bool = True) -> torch.Tensor:
        """softmax_fp32(q k^T * scale) v per head.  q: (B,Sq,H,64), k/v: (B,Sk,H,64) -> (B,Sq,H*64).

        causal_offset = number of cached positions before q's first row (query i sees keys <= offset+i).
        bf16 policy, dense=True (encoder / prefill / detokenizer: the engine's matrix-core attention): the probabilities
        that multiply V are rounded to bf16 while the normaliser sums them in fp32; dense=False (the single-query decode
        kernel) keeps them fp32."""
        q, k, v = self.rin(q), self.rin(k), self.rin(v)
        B, Sq, H, D = q.shape
        Sk = k.shape[1]
        kt = k.permute(0, 2, 3, 1)          # B,H,D,Sk
        vt = v.permute(0, 2, 1, 3)          # B,H,Sk,D
        out = torch.empty(B, Sq, H * D, device=q.device)
        for s in range(0, Sq, q_chunk):
            e = min(Sq, s + q_chunk)
            w = (q[:, s:e].permute(0, 2, 1, 3) @ kt) * scale       # B,H,c,Sk
            if causal_offset is not None:
                qi = torch.arange(s, e, device=q.device)[:, None] + causal_offset
                kj = torch.arange(Sk, device=q.device)[None, :]
                w = w.masked_fill(kj > qi, float("-inf"))
            if self.policy in ROUND16 and dense:
                w = w.float()
                pe = torch.exp(w - w.max(dim=-1, keepdim=True).values)
                o = (ROUND16[self.policy](pe) @ vt) / pe.sum(dim=-1, keepdim=True)
            else:
                o = torch.softmax(w.float(), dim=-1) @ vt
            out[:, s:e] = o.permute(0, 2, 1, 3).reshape(B, e - s, H * D)
        return out

    # ------------------------------------------------------------------ point encoder (miche)
    @_api
    def fourier_embed(self, pc: torch.Tensor) -> torch.Tensor:
        """FourierEmbedder.forward, embedder.py:87-105 with logspace freqs 2^0..2^(F-1), include_pi=False,
        include_input=True: cat(x, sin(x (x) f), cos(x (x) f)); (dim, freq) row-major inside sin/cos."""
        freqs = 2.0 ** torch.arange(self.cfg.num_freqs, dtype=torch.float32, device=pc.device)
        embed = (pc[..., None].contiguous() * freqs).view(*pc.shape[:-1], -1)
        return torch.cat((pc, embed.sin(), embed.cos()), dim=-1)

    def _miche_attn_block(self, x: torch.Tensor, p: str) -> torch.Tensor:
        """ResidualAttentionBlock._forward, transformer_blocks.py:109-112; MultiheadAttention 40-45;
        QKVMultiheadAttention 56-74 (per-head [q|k|v] split of the c_qkv output, scale 64^-1/4 on q and k)."""
        cfg = self.cfg
        B, n, Wd = x.shape
        qkv = self.linear(self.ln(x, p + "ln_1.", 1e-5), p + "attn.c_qkv.weight")
        qkv = qkv.view(B, n, cfg.enc_heads, 3 * 64)
        q, k, v = torch.split(qkv, 64, dim=-1)
        a = self.attention(q, k, v, scale=1.0 / math.sqrt(64))      # (q s)(k s) with s = 64^-1/4  ==  q k / 8
        x = x + self.linear(a, p + "attn.c_proj.weight", p + "attn.c_proj.bias")
        h = self.linear(self.ln(x, p + "ln_2.", 1e-5), p + "mlp.c_fc.weight", p + "mlp.c_fc.bias")
        x = x + self.linear(F.gelu(h), p + "mlp.c_proj.weight", p + "mlp.c_proj.bias")   # nn.GELU() = erf form
        return x

    @_api
    def encode_latents(self, pc_normal: torch.Tensor) -> torch.Tensor:
        """AlignedShapeAsLatentPLModule.encode_latents (asl_pl_module.py:145-157) ->
        AlignedShapeLatentPerceiver.encode_latents (sal_perceiver.py:372-381) ->
        CrossAttentionEncoder._forward (sal_perceiver.py:74-99).  (B,N,6) -> (B,T,W)."""
        with self._enc():
            return self._encode_latents(pc_normal)

    def _encode_latents(self, pc_normal: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        x = pc_normal.float()
        pc, feats = x[..., 0:3], x[..., 3:6]
        B = pc.shape[0]
        data = torch.cat([self.fourier_embed(pc), feats], dim=-1)
        data = self.linear(data, SM + "encoder.input_proj.weight", SM + "encoder.input_proj.bias")
        query = self.sd[SM + "encoder.query"][None].expand(B, -1, -1)
        # ResidualCrossAttentionBlock.forward, transformer_blocks.py:223-226; MultiheadCrossAttention 146-152;
        # QKVMultiheadCrossAttention 166-185 (kv viewed (B,N,heads,128) and split [k|v])
        p = SM + "encoder.cross_attn."
        q = self.linear(self.ln(query, p + "ln_1.", 1e-5), p + "attn.c_q.weight").view(B, -1, cfg.enc_heads, 64)
        kv = self.linear(self.ln(data, p + "ln_2.", 1e-5), p + "attn.c_kv.weight").view(B, -1, cfg.enc_heads, 128)
        k, v = torch.split(kv, 64, dim=-1)
        a = self.attention(q, k, v, scale=1.0 / math.sqrt(64))
        lat = query + self.linear(a, p + "attn.c_proj.weight", p + "attn.c_proj.bias")
        h = self.linear(self.ln(lat, p + "ln_3.", 1e-5), p + "mlp.c_fc.weight", p + "mlp.c_fc.bias")
        lat = lat + self.linear(F.gelu(h), p + "mlp.c_proj.weight", p + "mlp.c_proj.bias")
        for n in range(cfg.enc_layers):
            lat = self._miche_attn_block(lat, SM + f"encoder.self_attn.resblocks.{n}.")
        lat = self.ln(lat, SM + "encoder.ln_post.", 1e-5)
        assert lat.shape[1] == cfg.cond_length
        return lat      # cat([shape_embed[:,None], latents]) is the identity re-assembly of x[:,0], x[:,1:]

    @_api
    def to_shape_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """asl_pl_module.py:182-185: encode_kl_embed(sample_posterior=False) (sal_perceiver.py:383-396;
        DiagonalGaussianDistribution.mode = first half of pre_kl's output, distributions.py:34,69-70),
        then decode = post_kl + `transformer` blocks (sal_perceiver.py:273-275)."""
        E = self.cfg.embed_dim
        with self._enc():
            mean = self.linear(latents, SM + "pre_kl.weight", SM + "pre_kl.bias", rows=slice(0, E))
            x = self.linear(mean, SM + "post_kl.weight", SM + "post_kl.bias")
            for n in range(self.cfg.shape_layers):
                x = self._miche_attn_block(x, SM + f"transformer.resblocks.{n}.")
        return x

    @_api
    def process_point_feature(self, point_feature: torch.Tensor) -> torch.Tensor:
        """MeshAnything.process_point_feature, meshanything.py:125-132 -> (B,T,hidden) decoder prefix."""
        with self._enc():
            head = self.linear(point_feature[:, 0], "cond_head_proj.weight", "cond_head_proj.bias")
            shape_latents = self.to_shape_latents(point_feature[:, 1:])
            rest = self.linear(torch.cat([point_feature[:, 1:], shape_latents], dim=-1), "cond_proj.weight", "cond_proj.bias")
        return torch.cat([head[:, None], rest], dim=1)

    # ------------------------------------------------------------------ autoregressive decoder (ShapeOPT)
    @_api
    def embed_tokens(self, ids: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """Input embedding of decode step(s): ids (n,) = token fed at step t (n,) (t >= 1 = number of tokens
        generated so far).  shape_opt.py:237-245 (embed_with_vae), 448-460 (OPTFacePositionalEmbedding:
        slot = id if id<3 else ((t-2) mod 9)+3, attention_mask[:, cond:] has t ones), 326-328 (cond_embed[1]),
        359/364 + OPTLearnedPositionalEmbedding (row = (cond_length + t - 1) + 2)."""
        cfg = self.cfg
        special = ids < 3
        e = torch.zeros(ids.shape[0], cfg.hidden, device=self.device)
        if special.any():
            e[special] = self.sd[DEC + "extra_embeds.weight"][ids[special]]
        if (~special).any():
            code = self.sd[DEC + "quantize_codebooks"][0][ids[~special] - 3]
            e[~special] = self.linear(code, DEC + "input_layer.weight", DEC + "input_layer.bias")
        slot = torch.where(special, ids, torch.remainder(t - 2, cfg.face_per_token) + 3)
        e = e + self.sd[DEC + "token_embed_positions.weight"][slot]
        e = e + self.sd[DEC + "cond_embed.weight"][1]
        e = e + self.sd[DEC + "embed_positions.weight"][cfg.cond_length + t - 1 + 2]
        return e

    @_api
    def embed_prefix(self, prefix: torch.Tensor) -> torch.Tensor:
        """shape_opt.py:331-337 (cond_embed[0]) + embed_positions rows 2..T+1 (prefill)."""
        T = prefix.shape[1]
        return prefix + self.sd[DEC + "cond_embed.weight"][0] + self.sd[DEC + "embed_positions.weight"][2:2 + T]

    @_api
    def opt_layers(self, h: torch.Tensor, cache: Optional[List[Tuple[torch.Tensor, torch.Tensor]]]) -> torch.Tensor:
        """24x post-LN OPTDecoderLayer ([3p] transformers 4.39.3; container copy modeling_opt.py:202-253):
        a = softmax(q k^T / 8) v (causal); h = LN(h + Wo a + bo); h = LN(h + W2 relu(W1 h + b1) + b2); eps 1e-5.
        h: (B,S,H) rows appended after the `cache` contents; cache is updated in place when given."""
        cfg = self.cfg
        B, S, H = h.shape
        for n in range(cfg.layers):
            p = DEC + f"layers.{n}."
            q = self.linear(h, p + "self_attn.q_proj.weight", p + "self_attn.q_proj.bias").view(B, S, cfg.heads, 64)
            k = self.linear(h, p + "self_attn.k_proj.weight", p + "self_attn.k_proj.bias").view(B, S, cfg.heads, 64)
            v = self.linear(h, p + "self_attn.v_proj.weight", p + "self_attn.v_proj.bias").view(B, S, cfg.heads, 64)
            past = 0
            if cache is not None:
                if cache[n] is not None:
                    pk, pv = cache[n]
                    past = pk.shape[1]
                    k = torch.cat([pk, k], dim=1)
                    v = torch.cat([pv, v], dim=1)
                cache[n] = (k, v)
            # S == 1 with a cache = one decode step (fp32 probabilities); everything else is the dense prefill kernel
            a = self.attention(q, k, v, scale=0.125, causal_offset=past, dense=not (S == 1 and past > 0))
            h = self.ln(h + self.linear(a, p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias"),
                        p + "self_attn_layer_norm.", 1e-5)
            f = F.relu(self.linear(h, p + "fc1.weight", p + "fc1.bias"))
            h = self.ln(h + self.linear(f, p + "fc2.weight", p + "fc2.bias"), p + "final_layer_norm.", 1e-5)
        return h

    @_api
    def lm_head(self, h: torch.Tensor) -> torch.Tensor:
        return self.linear(h, "transformer.lm_head.weight")          # shape_opt.py:24,155 (no bias, un-tied)

    @staticmethod
    def topk_topp_filter(logits: torch.Tensor, top_k: int = 50, top_p: float = 0.95) -> Tuple[torch.Tensor, torch.Tensor]:
        """[3p] TopKLogitsWarper then TopPLogitsWarper (container copy
        transformers/generation/logits_process.py:473-560) for ONE row of logits.
        Returns (kept token ids, their final probabilities) ordered by descending logit, ties by ascending id.
        top-k keeps every score >= the k-th largest; top-p sorts ascending, removes the prefix whose cumulative
        softmax mass is <= 1 - top_p, always keeps the largest."""
        logits = logits.detach().cpu()                             # scalar walks below: never element-wise off a device
        V = logits.shape[0]
        k = min(top_k, V)
        order = sorted(range(V), key=lambda i: (-float(logits[i]), i))
        kth = logits[order[k - 1]]
        cand = [i for i in order if float(logits[i]) >= float(kth)]
        sc = logits[cand].float()                                  # descending
        p_desc = torch.softmax(sc, dim=0)
        # ascending cumulative mass: asc_cum[j] = sum of probs of candidates ranked j..last (descending index)
        asc = torch.flip(p_desc, dims=[0])
        asc_cum = torch.cumsum(asc, dim=0)
        remove_asc = asc_cum <= (1.0 - top_p)
        remove_asc[-1] = False                                      # min_tokens_to_keep = 1
        keep_desc = ~torch.flip(remove_asc, dims=[0])
        kept = [c for c, kp in zip(cand, keep_desc.tolist()) if kp]
        probs = torch.softmax(logits[kept].float(), dim=0)
        return torch.tensor(kept, dtype=torch.long), probs

    @staticmethod
    def sample_from(kept: torch.Tensor, probs: torch.Tensor, u: float) -> int:
        """Inverse-CDF draw over the kept tokens in descending-probability order with an injected uniform u
        (the reference uses torch.multinomial on the CUDA Philox stream, which cannot be reproduced across
        devices -- SURVEY.md section 3.5; parity is defined on injected uniforms)."""
        c = 0.0
        acc = torch.zeros((), dtype=torch.float32)
        for j in range(kept.shape[0]):
            acc = acc + probs[j]
            if float(acc) > u:
                return int(kept[j])
        return int(kept[-1])

    def pick(self, logits: torch.Tensor, sampling: bool, u: Optional[float], suppress_eos: bool) -> int:
        if suppress_eos:
            logits = logits.clone()
            logits[EOS] = float("-inf")
        if not sampling:
            return int(torch.argmax(logits))                        # lowest index wins ties
        kept, probs = self.topk_topp_filter(logits)
        return self.sample_from(kept, probs, float(u))

    @_api
    def generate(self, prefix: torch.Tensor, max_new_tokens: Optional[int] = None, sampling: bool = False,
                 uniforms: Optional[np.ndarray] = None, suppress_eos: bool = False,
                 return_logits: bool = False):
        """[3p] GenerationMixin.generate (greedy / sample), call site meshanything.py:143-162: prefill on
        inputs_embeds, then one token per step; a finished row keeps emitting pad=2; stop when every row has
        emitted eos=1 or after max_new_tokens.  Returns LongTensor (B, n_generated) (new tokens only)."""
        cfg = self.cfg
        B = prefix.shape[0]
        maxn = cfg.max_new_tokens if max_new_tokens is None else max_new_tokens
        out = torch.full((B, maxn), PAD, dtype=torch.long, device=self.device)
        all_logits = []
        n_done = 0
        for b in range(B):                      # rows are independent; run them one by one
            cache: List = [None] * cfg.layers
            h = self.opt_layers(self.embed_prefix(prefix[b:b + 1]), cache)
            logits = self.lm_head(h[0, -1])
            row_logits = [logits]
            tok = self.pick(logits, sampling, None if uniforms is None else uniforms[b, 0], suppress_eos)
            out[b, 0] = tok
            n = 1
            while n < maxn and tok != EOS:
                e = self.embed_tokens(torch.tensor([tok], device=self.device), torch.tensor([n], device=self.device))
                h = self.opt_layers(e[None], cache)
                logits = self.lm_head(h[0, -1])
                if return_logits:
                    row_logits.append(logits)
                tok = self.pick(logits, sampling, None if uniforms is None else uniforms[b, n], suppress_eos)
                out[b, n] = tok
                n += 1
            n_done = max(n_done, n)
            all_logits.append(row_logits)
        out = out[:, :n_done]
        return (out, all_logits) if return_logits else out

    @_api
    def teacher_forced_logits(self, prefix: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        """Logits the decoder assigns at every step when fed `tokens` (n,) as its own past output, computed as
        ONE causal pass (prefix + embedded tokens[:-1]) instead of n cached steps.  Row j = distribution of
        token j.  Mathematically identical to stepping (causal attention); used to verify long device streams
        quickly on CPU.  prefix: (1,T,H)."""
        n = tokens.shape[0]
        h0 = self.embed_prefix(prefix)
        if n > 1:
            e = self.embed_tokens(tokens[:-1], torch.arange(1, n, device=self.device))
            h0 = torch.cat([h0, e[None]], dim=1)
        h = self.opt_layers(h0, None)
        return self.lm_head(h[0, self.cfg.cond_length - 1:])

    # ------------------------------------------------------------------ post-processing + detokenizer
    @_api
    def postprocess_tokens(self, results: torch.Tensor) -> torch.Tensor:
        """meshanything.py:141-142,163-172: pad to generate_length with eos, drop first and last slot,
        {bos,eos,pad} -> -1, others -= 3.  (B, <=max_new) -> (B, n_max_faces*9) in [-1, codebook_size)."""
        B = results.shape[0]
        L = self.cfg.max_new_tokens
        assert results.shape[1] <= L
        outputs = torch.ones(B, L, dtype=torch.long, device=results.device) * EOS
        outputs[:, :results.shape[1]] = results
        outputs = outputs[:, 1:-1].clone()
        outputs[outputs == BOS] = PAD_ID
        outputs[outputs == EOS] = PAD_ID
        outputs[outputs == PAD] = PAD_ID
        outputs[outputs != PAD_ID] -= 3
        return outputs

    @_api
    def get_codes(self, indices: torch.Tensor) -> torch.Tensor:
        """MeshAnything.get_codes, meshanything.py:178-212: per vertex, sum of the 3 (shared-codebook) rows;
        pad (-1) entries contribute zero.  (B, nf*9) -> (B, nf*3, codebook_dim)."""
        B = indices.shape[0]
        idx = indices.reshape(B, -1, 3)
        mask = idx == PAD_ID
        codes = self.sd[DEC + "quantize_codebooks"][0][idx.masked_fill(mask, 0)]      # B,n,3,D
        codes = codes.masked_fill(mask[..., None], 0.0)
        # reduce(codes, 'q ... -> ...', 'sum'): einops sums the quantizer axis (q = 0,1,2 in order)
        return codes[:, :, 0] + codes[:, :, 1] + codes[:, :, 2]

    def _bert_layer(self, x: torch.Tensor, p: str) -> torch.Tensor:
        """[3p] BERT post-LN encoder layer (bert-base shape, eps 1e-12, GELU erf), no attention mask
        (meshanything.py:62-64 passes none)."""
        cfg = self.cfg
        B, S, Wd = x.shape
        q = self.linear(x, p + "attention.self.query.weight", p + "attention.self.query.bias").view(B, S, cfg.tok_heads, 64)
        k = self.linear(x, p + "attention.self.key.weight", p + "attention.self.key.bias").view(B, S, cfg.tok_heads, 64)
        v = self.linear(x, p + "attention.self.value.weight", p + "attention.self.value.bias").view(B, S, cfg.tok_heads, 64)
        a = self.attention(q, k, v, scale=0.125)
        x = self.ln(x + self.linear(a, p + "attention.output.dense.weight", p + "attention.output.dense.bias"),
                    p + "attention.output.LayerNorm.", 1e-12)
        f = F.gelu(self.linear(x, p + "intermediate.dense.weight", p + "intermediate.dense.bias"))
        x = self.ln(x + self.linear(f, p + "output.dense.weight", p + "output.dense.bias"), p + "output.LayerNorm.", 1e-12)
        return x

    @_api
    def detok_point_feature(self, encode_feature: torch.Tensor) -> torch.Tensor:
        """NoiseResistantDecoder.process_point_feature, meshanything.py:42-48."""
        with self._enc():
            head = self.linear(encode_feature[:, 0], TOK + "cond_head_proj.weight", TOK + "cond_head_proj.bias")
            rest = self.linear(encode_feature[:, 1:], TOK + "cond_proj.weight", TOK + "cond_proj.bias")
        pf = torch.cat([head[:, None], rest], dim=1)
        return self.ln(pf + self.sd[TOK + "point_pe.weight"][None, :pf.shape[1]], TOK + "point_layernorm.", 1e-5)

    @_api
    def detokenize(self, input_ids: torch.Tensor, input_embeds: torch.Tensor, point_feature: torch.Tensor,
                   return_logits: bool = False):
        """NoiseResistantDecoder.forward, meshanything.py:50-80.  ids (B,nf*9) in [-1,C), embeds (B,nf*3,D),
        point_feature (B,T,W) raw encoder latents -> (B,nf,3,3) fp32 with NaN rows for invalid faces."""
        cfg = self.cfg
        B = input_ids.shape[0]
        input_ids = input_ids.reshape(B, -1)
        pf = self.detok_point_feature(point_feature)
        nf = input_embeds.shape[1] // 3
        face_embeds = input_embeds.reshape(B, nf, 3 * input_embeds.shape[2])
        face_embeds = self.linear(face_embeds, TOK + "project_down_codebook.weight", TOK + "project_down_codebook.bias")
        face_mask = (input_ids != PAD_ID).reshape(B, nf, 9).all(dim=-1)
        face_embeds = face_embeds.masked_fill(~face_mask[..., None], 0.0)
        face_embeds = self.ln(face_embeds + self.sd[TOK + "pos_embedding.weight"][None, :nf], TOK + "layernorm.", 1e-5)
        x = torch.cat([pf, face_embeds], dim=1)
        for n in range(cfg.tok_layers):
            x = self._bert_layer(x, TOK + f"decoder.layer.{n}.")
        decoded = x[:, cfg.cond_length:]
        decoded = decoded.masked_fill(~face_mask[..., None], 0.0)
        logits = self.linear(decoded, TOK + "to_coor_logits.0.weight", TOK + "to_coor_logits.0.bias")
        logits = logits.reshape(B, nf, 9, cfg.discrete_num)
        coords = logits.argmax(dim=-1).reshape(B, nf, 3, 3)
        cont = undiscretize(coords, low=-0.5, high=0.5, num_discrete=cfg.discrete_num)
        cont = cont.masked_fill(~face_mask[:, :, None, None], float("nan"))
        return (cont, logits) if return_logits else cont

    # ------------------------------------------------------------------ facade
    @_api
    def forward(self, pc_normal: torch.Tensor, sampling: bool = False, uniforms: Optional[np.ndarray] = None,
                max_new_tokens: Optional[int] = None, suppress_eos: bool = False) -> Dict[str, torch.Tensor]:
        """MeshAnything.forward, meshanything.py:134-176."""
        point_feature = self.encode_latents(pc_normal)
        prefix = self.process_point_feature(point_feature)
        results = self.generate(prefix, max_new_tokens, sampling, uniforms, suppress_eos)
        ids = self.postprocess_tokens(results)
        code_embed = self.get_codes(ids)
        coords = self.detokenize(ids, code_embed, point_feature)
        return {"point_feature": point_feature, "prefix": prefix, "tokens": results, "ids": ids, "coords": coords}


def verify_greedy_stream(oracle: Oracle, prefix: torch.Tensor, tokens: torch.Tensor, tol: float,
                         suppress_eos: bool = False) -> Dict[str, object]:
    """Check that `tokens` (n,) is a valid greedy decode of the oracle model for `prefix` (1,T,H).

    Teacher-forces the stream (one causal pass) and, at every step, compares the oracle's argmax with the
    stream's token.  A disagreement whose oracle logit margin (top1 - logit[stream token]) is <= tol is an
    *ambiguous step* (the two implementations differ by summation order / a bf16 rounding flip and the step
    was a near-tie); a larger margin is a hard mismatch.  Returns counts and the worst margin."""
    with oracle.on_device():
        logits = oracle.teacher_forced_logits(prefix, tokens)
    tokens = tokens.to(logits.device)
    if suppress_eos:
        logits = logits.clone()
        logits[:, EOS] = float("-inf")
    top = logits.argmax(dim=-1)
    n = tokens.shape[0]
    diff = (top[:n] != tokens).nonzero().flatten().tolist()
    margins = [float(logits[j, top[j]] - logits[j, tokens[j]]) for j in diff]
    hard = [(j, m) for j, m in zip(diff, margins) if not (m <= tol)]
    srt = torch.sort(logits[:n].float(), dim=-1, descending=True).values
    gaps = (srt[:, 0] - srt[:, 1])
    return {"n": n, "ambiguous": len(diff) - len(hard), "hard": hard, "worst_margin": max(margins) if margins else 0.0,
            "median_top_gap": float(gaps.median()), "min_top_gap": float(gaps.min())}


def verify_sampled_stream(oracle: Oracle, prefix: torch.Tensor, tokens: torch.Tensor, uniforms: np.ndarray, tol: float = 1e-4,
                          suppress_eos: bool = False) -> Dict[str, object]:
    """Teacher-forced check of a top-k/top-p sampled stream drawn with injected uniforms: at every step the
    oracle's own filtered distribution must put `tokens[j]` on the CDF interval that contains uniforms[j]
    (within `tol`: two implementations differ by fp32 summation order in the logits)."""
    logits = oracle.teacher_forced_logits(prefix, tokens).cpu()
    tokens = tokens.cpu()
    n = tokens.shape[0]
    exact, ambiguous, hard = 0, 0, []
    for j in range(n):
        lg = logits[j].clone()
        if suppress_eos:
            lg[EOS] = float("-inf")
        kept, probs = Oracle.topk_topp_filter(lg)
        pick = Oracle.sample_from(kept, probs, float(uniforms[j]))
        tok = int(tokens[j])
        if pick == tok:
            exact += 1
            continue
        kl = kept.tolist()
        ok = False
        if tok in kl:
            c = np.concatenate([[0.0], np.cumsum(probs.double().numpy())])
            i = kl.index(tok)
            ok = (c[i] - tol) <= float(uniforms[j]) <= (c[i + 1] + tol)
        if not ok:
            # The kept SET can differ by one token when the ascending cumulative mass at the top-p cut sits within `tol`
            # of 1 - top_p: re-draw with one token more / fewer kept, but only if that boundary really is that close.
            V = lg.shape[0]
            order = sorted(range(V), key=lambda q: (-float(lg[q]), q))[:min(50, V)]
            pk = torch.softmax(lg[order].double(), dim=0).numpy()                # over the top-k candidates, descending
            tail = np.cumsum(pk[::-1])[::-1]                                     # tail[r] = mass of candidates ranked r..last
            nkept = len(kl)
            for nk in (nkept - 1, nkept + 1):
                if nk < 1 or nk > len(order) or tok not in order[:nk]:
                    continue
                r = nkept if nk > nkept else nkept - 1                           # rank of the token that changes sides
                if abs(float(tail[r]) - 0.05) > tol:
                    continue
                alt = order[:nk]
                pa = torch.softmax(lg[alt].double(), dim=0).numpy()
                c = np.concatenate([[0.0], np.cumsum(pa)])
                i = alt.index(tok)
                if (c[i] - tol) <= float(uniforms[j]) <= (c[i + 1] + tol):
                    ok = True
                    break
        if ok:
            ambiguous += 1
        else:
            hard.append((j, tok, pick))
    return {"n": n, "exact": exact, "ambiguous": ambiguous, "hard": hard}


def classify_sampled_draws(logits: torch.Tensor, tokens: torch.Tensor, uniforms: torch.Tensor, tol: float, top_k: int = 50,
                           top_p: float = 0.95) -> Dict[str, torch.Tensor]:
    """`verify_sampled_stream`'s per-step decision for MANY steps at once, as tensor operations on `logits.device` (the scalar walk
    costs a Python sort of the vocabulary per step: minutes for 64 rows x 100 steps at the 350M shape).  logits (N, V) with eos already
    suppressed if it has to be; tokens (N,), uniforms (N,).  Same definitions: top-k keeps the k largest scores (descending, ties by
    torch.topk's order -- a tie AT the k-th score is the one case the scalar walk treats differently: it keeps both); top-p removes the
    ascending prefix whose cumulative mass is <= 1 - top_p and always keeps the largest; inverse CDF in descending order.
    Returns boolean vectors `exact` (the oracle's own draw is the token), `ok` (the token's CDF interval, or the one with a single
    candidate more / fewer kept when the top-p cut sits within tol of the boundary, contains the uniform within tol), and `distance`:
    how far the uniform lies from the token's interval on the oracle's CDF (0 inside; a candidate the oracle's top-p cut removed, or one of
    the 14 scores below the k-th, sits at the end of the CDF: distance 1 - u; any other token: inf).  The distance is the robust statistic when the
    distribution is nearly flat (random-init weights: the top-50 logits span ~1.5, neighbours ~0.03 apart, so bf16 noise reorders them
    and moves the top-p cut by more than one candidate)."""
    N, V = logits.shape
    k = min(top_k, V)
    lg = logits.float()
    tokens = tokens.to(lg.device).long()
    u = uniforms.to(lg.device).float()
    k_ext = min(V, k + 14)                                                                 # neighbourhood of the k-th score (see `distance`)
    topv_e, topi_e = torch.topk(lg, k_ext, dim=-1, sorted=True)
    topv, topi = topv_e[:, :k], topi_e[:, :k]
    p_desc = torch.softmax(topv, dim=-1)
    tail = torch.flip(torch.cumsum(torch.flip(p_desc, dims=[-1]), dim=-1), dims=[-1])      # tail[r] = mass of ranks r..k-1
    keep = ~(tail <= (1.0 - top_p))
    keep[:, 0] = True
    nk = keep.sum(dim=-1)                                                                  # kept ranks are a prefix 0..nk-1
    rank_ar = torch.arange(k, device=lg.device)[None, :]
    is_tok = topi_e == tokens[:, None]
    near_topk = is_tok.any(dim=-1)                                                         # among the k + 14 largest scores
    tok_rank = torch.where(near_topk, is_tok.float().argmax(dim=-1), torch.full_like(nk, k_ext))

    def interval(nkeep):
        mask = rank_ar < nkeep[:, None]
        probs = torch.softmax(torch.where(mask, topv, torch.full_like(topv, float("-inf"))), dim=-1)
        # sample_from accumulates in fp32, one candidate at a time: a sequential cumsum
        cdf = torch.cumsum(probs, dim=-1)
        lo = torch.cat([torch.zeros(N, 1, device=lg.device), cdf[:, :-1]], dim=-1)
        r = tok_rank.clamp(max=k - 1)
        t_lo, t_hi = lo.gather(1, r[:, None])[:, 0], cdf.gather(1, r[:, None])[:, 0]
        kept_tok = tok_rank < nkeep
        # the oracle's own draw: first rank whose cumulative mass exceeds u, else the last kept one
        over = (cdf > u[:, None]) & mask
        first = torch.where(over.any(dim=-1), over.float().argmax(dim=-1), nkeep - 1)
        return kept_tok, t_lo, t_hi, first

    kept_tok, t_lo, t_hi, first = interval(nk)
    exact = topi.gather(1, first[:, None])[:, 0] == tokens
    ok = exact | (kept_tok & (t_lo - tol <= u) & (u <= t_hi + tol))
    distance = torch.where(kept_tok, torch.maximum(t_lo - u, u - t_hi).clamp(min=0.0), 1.0 - u)
    distance = torch.where(near_topk, distance, torch.full_like(distance, float("inf")))
    for delta in (-1, 1):
        nk_alt = nk + delta
        valid = (nk_alt >= 1) & (nk_alt <= k)
        r = torch.where(torch.full_like(nk, delta) > 0, nk, nk - 1).clamp(0, k - 1)      # rank of the candidate that changes sides
        near = (tail.gather(1, r[:, None])[:, 0] - (1.0 - top_p)).abs() <= tol
        kt, lo2, hi2, _ = interval(nk_alt.clamp(1, k))
        ok = ok | (valid & near & kt & (lo2 - tol <= u) & (u <= hi2 + tol))
    return {"exact": exact, "ok": ok, "distance": distance}


def verify_sampled_batch(oracle: Oracle, prefix: torch.Tensor, tokens: torch.Tensor, uniforms: torch.Tensor, tol: float,
                         suppress_eos: bool = False) -> Dict[str, object]:
    """`verify_sampled_stream` for a batch: prefix (B, T, H), tokens (B, n), uniforms (B, n).  One teacher-forced pass per row on the
    oracle's device, the per-draw decisions vectorised (classify_sampled_draws)."""
    B, n = tokens.shape
    rows = []
    with oracle.on_device():
        for b in range(B):
            lg = oracle.teacher_forced_logits(prefix[b:b + 1], tokens[b])[:n]
            if suppress_eos:
                lg = lg.clone()
                lg[:, EOS] = float("-inf")
            rows.append(lg)
    logits = torch.cat(rows)
    c = classify_sampled_draws(logits, tokens.reshape(-1), uniforms.reshape(-1), tol)
    exact, ok = c["exact"].cpu(), c["ok"].cpu()
    bad = (~ok).nonzero().flatten().tolist()
    return {"n": B * n, "exact": int(exact.sum()), "ambiguous": int((ok & ~exact).sum()), "hard": [(i // n, i % n) for i in bad],
            "distance": c["distance"].cpu().reshape(B, n)}
