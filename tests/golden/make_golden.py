#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE'S OWN CODE (imported from /root/reference) and the
container's transformers copy on the seeded synthetic checkpoint (meshanything_amd/checkpoint.py).

Runs only in the authoring container (needs /root/reference); the fixtures it writes are committed and
travel to the GPU box.  Nothing here is imported by the product.  Recipe: SURVEY.md Appendix B --
stub the import-time-only modules that are not installed (omegaconf, trimesh, skimage, cv2, mesh2sdf),
patch `to_bettertransformer` to the identity (vanilla BertLayer = same math) and make
`AutoConfig.from_pretrained("bert-base-uncased")` return `BertConfig()` (its defaults are bert-base).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from meshanything_amd.config import MAConfig            # noqa: E402
from meshanything_amd.checkpoint import synthetic_state_dict, PE, SM, DEC, TOK  # noqa: E402


# the anchors of full_anchor_hf.npz: (tag, seed, init of synthetic_state_dict, how the token path is chosen)
ANCHOR_HF_SETS = (("dva", 1234, "diverse", "greedy"), ("hfa", 1234, "hf", "sampled"))


def _stub_modules():
    for name in ("omegaconf", "trimesh", "skimage", "skimage.measure", "cv2", "mesh2sdf", "mesh2sdf.core"):
        try:
            __import__(name)
        except Exception:
            m = types.ModuleType(name)
            if name == "omegaconf":
                m.OmegaConf = type("OmegaConf", (), {})
                m.DictConfig = dict
            sys.modules[name] = m
    import transformers
    transformers.PreTrainedModel.to_bettertransformer = lambda self: self
    _orig = transformers.AutoConfig.from_pretrained

    def _from_pretrained(name, *a, **k):
        if name == "bert-base-uncased":
            return transformers.BertConfig()
        return _orig(name, *a, **k)
    transformers.AutoConfig.from_pretrained = staticmethod(_from_pretrained)


def build_perceiver(cfg: MAConfig, sd):
    from MeshAnything.miche.michelangelo.models.tsal.sal_perceiver import AlignedShapeLatentPerceiver
    from MeshAnything.miche.michelangelo.models.tsal.clip_asl_module import CLIPAlignedShapeAsLatentModule
    shape_model = AlignedShapeLatentPerceiver(
        device=None, dtype=None, num_latents=cfg.num_latents, embed_dim=cfg.embed_dim, point_feats=3,
        num_freqs=cfg.num_freqs, include_pi=False, heads=cfg.enc_heads, width=cfg.enc_width,
        num_encoder_layers=cfg.enc_layers, num_decoder_layers=cfg.shape_layers, use_ln_post=True,
        init_scale=0.25, qkv_bias=False, use_checkpoint=True)
    model = CLIPAlignedShapeAsLatentModule(shape_model=shape_model)
    sub = {k[len(PE):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(PE)}
    missing, unexpected = model.load_state_dict(sub, strict=True), None
    return model.eval()


class RefPointEncoder:
    """The 12 lines of AlignedShapeAsLatentPLModule.encode_latents / to_shape_latents (asl_pl_module.py:145-157,
    182-185) driven on the reference's own perceiver (the PL module itself needs omegaconf/pytorch_lightning)."""
    def __init__(self, model):
        self.model = model

    @torch.no_grad()
    def encode_latents(self, surface):
        pc = surface[..., 0:3]
        feats = surface[..., 3:6]
        shape_embed, shape_latents = self.model.shape_model.encode_latents(pc=pc, feats=feats)
        shape_embed = shape_embed.unsqueeze(1)
        return torch.cat([shape_embed, shape_latents], dim=1)

    @torch.no_grad()
    def to_shape_latents(self, latents):
        shape_zq, posterior = self.model.shape_model.encode_kl_embed(latents, sample_posterior=False)
        return self.model.shape_model.decode(shape_zq)


def synth_cloud(seed: int, n: int) -> np.ndarray:
    """SURVEY.md section 8d cfg 3: unit-sphere directions * U(0.3,1) radius, normals = directions (pre-normalisation)."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    r = 0.3 + 0.7 * torch.rand(n, 1, generator=g)
    return torch.cat([d * r, d], dim=-1).numpy().astype(np.float32)


def golden_dataset(out):
    import main as ref_main                                            # /root/reference/main.py
    src = np.load(os.path.join(REF, "pc_examples/mouse.npy"))
    np.random.seed(0)                                                  # accelerate.set_seed(0) -> np.random.seed(0)
    ds = ref_main.Dataset("pc_normal", [os.path.join(REF, "pc_examples/mouse.npy")])
    item = ds[0]["pc_normal"]
    out["mouse_raw"] = src
    out["mouse_norm"] = item
    out["mouse_norm_sha256"] = np.frombuffer(hashlib.sha256(item.tobytes()).digest(), dtype=np.uint8)
    # a float32 cloud through the same normalisation
    cloud = synth_cloud(7, 5000)
    np.random.seed(3)
    idx = np.random.choice(cloud.shape[0], 4096, replace=False)
    ds.data = [{"pc_normal": cloud[idx], "uid": "synth"}]
    out["synth_raw"] = cloud
    out["synth_norm"] = ds[0]["pc_normal"]
    print("dataset: mouse sha256", hashlib.sha256(item.tobytes()).hexdigest())


def golden_encoder(out, cfg: MAConfig, sd, tag: str, pc_normal: np.ndarray, rows):
    from MeshAnything.models.meshanything import MeshAnything as RefMeshAnything
    enc = RefPointEncoder(build_perceiver(cfg, sd))
    x = torch.from_numpy(pc_normal.astype(np.float32))[None]
    lat = enc.encode_latents(x)
    shp = enc.to_shape_latents(lat[:, 1:])
    ns = types.SimpleNamespace()
    ns.cond_length = cfg.cond_length
    ns.config = types.SimpleNamespace(word_embed_proj_dim=cfg.hidden)
    ns.cond_head_proj = torch.nn.Linear(cfg.enc_width, cfg.hidden)
    ns.cond_proj = torch.nn.Linear(2 * cfg.enc_width, cfg.hidden)
    ns.cond_head_proj.load_state_dict({"weight": torch.from_numpy(sd["cond_head_proj.weight"]), "bias": torch.from_numpy(sd["cond_head_proj.bias"])})
    ns.cond_proj.load_state_dict({"weight": torch.from_numpy(sd["cond_proj.weight"]), "bias": torch.from_numpy(sd["cond_proj.bias"])})
    ns.point_encoder = enc
    with torch.no_grad():
        prefix = RefMeshAnything.process_point_feature(ns, lat)          # meshanything.py:125-132, unbound
    from MeshAnything.miche.michelangelo.models.modules.embedder import FourierEmbedder
    four = FourierEmbedder(num_freqs=cfg.num_freqs, include_pi=False)(x[0, :16, :3])
    out[f"{tag}_input"] = pc_normal
    out[f"{tag}_fourier16"] = four.numpy()
    rows = np.asarray(rows)
    out[f"{tag}_rows"] = rows
    out[f"{tag}_latents_rows"] = lat[0, rows].numpy()
    out[f"{tag}_latents_cols8"] = lat[0, :, :8].numpy()
    out[f"{tag}_shape_rows"] = shp[0, rows[rows < cfg.num_latents]].numpy()
    out[f"{tag}_shape_cols8"] = shp[0, :, :8].numpy()
    out[f"{tag}_prefix_rows"] = prefix[0, rows].numpy()
    out[f"{tag}_prefix_cols8"] = prefix[0, :, :8].numpy()
    out[f"{tag}_stats"] = np.array([lat.double().sum(), lat.double().abs().sum(), shp.double().sum(), shp.double().abs().sum(),
                                    prefix.double().sum(), prefix.double().abs().sum()])
    print(f"encoder[{tag}]: latents absmax {lat.abs().max():.3f} shape absmax {shp.abs().max():.3f} prefix absmax {prefix.abs().max():.3f}")
    return lat, prefix


def golden_decoder_embed(out, cfg: MAConfig, sd):
    """embed_with_vae (shape_opt.py:237-245), OPTFacePositionalEmbedding (448-460), cond_embed (326-328),
    OPTLearnedPositionalEmbedding ([3p], container copy) driven exactly as ShapeOPTDecoder.forward 318-364 does."""
    from MeshAnything.models.shape_opt import ShapeOPTConfig, ShapeOPTDecoder
    c = ShapeOPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=1, ffn_dim=cfg.ffn,
                       num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_positions,
                       do_layer_norm_before=False, word_embed_proj_dim=cfg.hidden, activation_function="relu",
                       bos_token_id=0, eos_token_id=1, pad_token_id=2)
    c.quantize_codebook_dim = cfg.codebook_dim
    c.face_per_token = 9
    c.cond_length = cfg.cond_length
    dec = ShapeOPTDecoder(c).eval()
    for nm in ("extra_embeds.weight", "input_layer.weight", "input_layer.bias", "embed_positions.weight",
               "token_embed_positions.weight", "cond_embed.weight"):
        obj = dec
        parts = nm.split(".")
        for p_ in parts[:-1]:
            obj = getattr(obj, p_)
        getattr(obj, parts[-1]).data.copy_(torch.from_numpy(sd[DEC + nm]))
    dec.quantize_codebooks = torch.nn.Parameter(torch.from_numpy(sd[DEC + "quantize_codebooks"]))
    T = cfg.cond_length
    ts = [1, 2, 3, 4, 10, 11, 12, 20, cfg.max_new_tokens - 1]
    toks = [0, 5, 3, cfg.vocab - 1, 7, 2, 9, 1, 33]
    es = []
    with torch.no_grad():
        for t, tok in zip(ts, toks):
            input_ids = torch.tensor([[tok]])
            attention_mask = torch.ones(1, T + t, dtype=torch.long)      # generate(): prefix + t generated so far
            e = dec.embed_with_vae(input_ids)
            e = e + dec.token_embed_positions(attention_mask[:, dec.cond_length:], None, input_ids, dec.face_per_token)
            e = e + dec.cond_embed(torch.ones(1, 1).long())
            pos = dec.embed_positions(attention_mask, T + t - 1)
            es.append((e + pos)[0, 0].numpy())
        # prefill: inputs_embeds + cond_embed[0] + positions
        pre = torch.from_numpy(np.arange(T * cfg.hidden, dtype=np.float32).reshape(1, T, cfg.hidden) * 1e-3)
        h0 = pre + dec.cond_embed(torch.zeros(1, T).long()) + dec.embed_positions(torch.ones(1, T, dtype=torch.long), 0)
    out["dec_embed_t"] = np.array(ts)
    out["dec_embed_tok"] = np.array(toks)
    out["dec_embed_e"] = np.stack(es)
    out["dec_prefill_h0_rows"] = h0[0, [0, 1, T - 1]].numpy()
    print("decoder embed: done")


def golden_opt_layers(out, cfg: MAConfig, sd):
    """[3p] OPTDecoderLayer (post-LN) x cfg.layers, causal, eager attention -- the container's transformers copy."""
    from transformers import OPTConfig
    from transformers.models.opt.modeling_opt import OPTDecoderLayer
    c = OPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, ffn_dim=cfg.ffn,
                  num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_positions,
                  do_layer_norm_before=False, word_embed_proj_dim=cfg.hidden, activation_function="relu")
    c._attn_implementation = "eager"
    layers = [OPTDecoderLayer(c, layer_idx=i).eval() for i in range(cfg.layers)]
    for i, L in enumerate(layers):
        sub = {k[len(DEC + f"layers.{i}."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(DEC + f"layers.{i}.")}
        L.load_state_dict(sub, strict=True)
    S = cfg.cond_length + 6
    g = torch.Generator().manual_seed(11)
    h = torch.randn(1, S, cfg.hidden, generator=g)
    mask = torch.full((S, S), float("-inf")).triu(1)[None, None]
    x = h
    with torch.no_grad():
        for L in layers:
            x = L(x, attention_mask=mask)
            if isinstance(x, tuple):
                x = x[0]
    out["opt_h_in"] = h.numpy()
    out["opt_h_out"] = x.numpy()
    print("opt layers: out absmax", float(x.abs().max()))


class _TupleCacheShim:
    """Duck-typed stand-in for transformers' Cache inside ONE layer call: holds the 4.39.3-style (k, v) tuple of that layer."""
    def __init__(self, past):
        self.past, self.present = past, None

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        if self.past is not None:
            key_states = torch.cat([self.past[0], key_states], dim=2)
            value_states = torch.cat([self.past[1], value_states], dim=2)
        self.present = (key_states, value_states)
        return key_states, value_states


class _Layer439(torch.nn.Module):
    """Adapter: the container's real OPTDecoderLayer (eager attention) behind the transformers==4.39.3 call signature that
    ShapeOPTDecoder.forward uses (shape_opt.py:403-415): returns (hidden_states, present_key_value), tuple KV cache of shape
    (B, heads, L, 64), causal attention (the flash-attn path of 4.39.3 gets attention_mask=None = "causal, no padding")."""
    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, hidden_states, attention_mask=None, layer_head_mask=None, past_key_value=None, output_attentions=False, use_cache=False):
        assert attention_mask is None and not output_attentions           # no padding anywhere in generate() on this path
        B, S, _ = hidden_states.shape
        past = 0 if past_key_value is None else past_key_value[0].shape[2]
        mask = torch.full((S, past + S), float("-inf")).triu(past + 1)[None, None]
        shim = _TupleCacheShim(past_key_value)
        out = self.layer(hidden_states, attention_mask=mask, past_key_values=shim)
        if isinstance(out, tuple):
            out = out[0]
        return (out, shim.present) if use_cache else (out,)


def golden_shapeopt_forward(out, cfg: MAConfig, sd):
    """The reference's OWN ShapeOPTDecoder.forward (shape_opt.py:248-438) driven like generate() drives it: one call with
    inputs_embeds = the prefix (attention_mask of T ones), then decode steps with input_ids (B,1), the returned tuple cache and
    an attention mask of T+t ones.  Its layers are the container's real OPTDecoderLayer behind a 4.39.3-signature adapter; the
    un-tied lm_head (shape_opt.py:24,155) is applied to every returned hidden state."""
    from transformers import OPTConfig
    from transformers.models.opt.modeling_opt import OPTDecoderLayer
    from MeshAnything.models.shape_opt import ShapeOPTConfig, ShapeOPTDecoder
    c = ShapeOPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, ffn_dim=cfg.ffn,
                       num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_positions,
                       do_layer_norm_before=False, word_embed_proj_dim=cfg.hidden, activation_function="relu",
                       bos_token_id=0, eos_token_id=1, pad_token_id=2)
    c.quantize_codebook_dim = cfg.codebook_dim
    c.face_per_token = 9
    c.cond_length = cfg.cond_length
    c._attn_implementation = "eager"
    dec = ShapeOPTDecoder(c).eval()
    dec._use_flash_attention_2 = True                     # the only branch the reference allows (shape_opt.py:347-357); masks stay 2-D
    for nm in ("extra_embeds.weight", "input_layer.weight", "input_layer.bias", "embed_positions.weight",
               "token_embed_positions.weight", "cond_embed.weight"):
        obj = dec
        parts = nm.split(".")
        for p_ in parts[:-1]:
            obj = getattr(obj, p_)
        getattr(obj, parts[-1]).data.copy_(torch.from_numpy(sd[DEC + nm]))
    dec.quantize_codebooks = torch.nn.Parameter(torch.from_numpy(sd[DEC + "quantize_codebooks"]))
    oc = OPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, ffn_dim=cfg.ffn,
                   num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_positions,
                   do_layer_norm_before=False, word_embed_proj_dim=cfg.hidden, activation_function="relu")
    oc._attn_implementation = "eager"
    layers = []
    for i in range(cfg.layers):
        L = OPTDecoderLayer(oc, layer_idx=i).eval()
        sub = {k[len(DEC + f"layers.{i}."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(DEC + f"layers.{i}.")}
        L.load_state_dict(sub, strict=True)
        layers.append(_Layer439(L))
    dec.layers = torch.nn.ModuleList(layers)
    lm_head = torch.from_numpy(sd["transformer.lm_head.weight"])
    B, T = 2, cfg.cond_length
    g = torch.Generator().manual_seed(31)
    prefix = torch.randn(B, T, cfg.hidden, generator=g) * 0.7
    # teacher-forced tokens: specials mid-sequence, both rows different, more than one face (slot cycle wraps)
    steps = 13
    toks = torch.randint(3, cfg.vocab, (B, steps), generator=g)
    toks[0, 0] = 0; toks[1, 0] = 0                        # the first generated token is expected to be bos
    toks[0, 5] = 2; toks[1, 7] = 1
    hs, logits = [], []
    with torch.no_grad():
        o = dec(inputs_embeds=prefix, attention_mask=torch.ones(B, T, dtype=torch.long), use_cache=True, return_dict=True)
        pkv = o.past_key_values
        hs.append(o.last_hidden_state[:, -1])
        for t in range(1, steps + 1):                     # step t feeds token t-1; the mask covers the prefix and t generated tokens
            o = dec(input_ids=toks[:, t - 1:t], past_key_values=pkv, attention_mask=torch.ones(B, T + t, dtype=torch.long),
                    use_cache=True, return_dict=True)
            pkv = o.past_key_values
            hs.append(o.last_hidden_state[:, -1])
        for h in hs:
            logits.append(h @ lm_head.T)
    out["sopt_prefix"] = prefix.numpy()
    out["sopt_tokens"] = toks.numpy()
    out["sopt_hidden"] = torch.stack(hs, dim=1).numpy()            # (B, steps + 1, H): after the prefill, then after each step
    out["sopt_logits_cols"] = torch.stack(logits, dim=1)[:, :, :48].numpy()
    out["sopt_logits_argmax"] = torch.stack(logits, dim=1).argmax(-1).numpy()
    out["sopt_cache_len"] = np.array([pkv[0][0].shape[2]])
    print("ShapeOPTDecoder.forward: hidden absmax", float(torch.stack(hs).abs().max()), "cache length", int(pkv[0][0].shape[2]))


def build_ref_decoder(cfg: MAConfig, sd):
    """The reference's own ShapeOPTDecoder (shape_opt.py:181-438) carrying the checkpoint, its layers = the container's real
    OPTDecoderLayer behind the 4.39.3-signature adapter (as in golden_shapeopt_forward); returns (decoder, lm_head weight)."""
    from transformers import OPTConfig
    from transformers.models.opt.modeling_opt import OPTDecoderLayer
    from MeshAnything.models.shape_opt import ShapeOPTConfig, ShapeOPTDecoder
    c = ShapeOPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, ffn_dim=cfg.ffn,
                       num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_positions,
                       do_layer_norm_before=False, word_embed_proj_dim=cfg.hidden, activation_function="relu",
                       bos_token_id=0, eos_token_id=1, pad_token_id=2)
    c.quantize_codebook_dim = cfg.codebook_dim
    c.face_per_token = 9
    c.cond_length = cfg.cond_length
    c._attn_implementation = "eager"
    dec = ShapeOPTDecoder(c).eval()
    dec._use_flash_attention_2 = True
    for nm in ("extra_embeds.weight", "input_layer.weight", "input_layer.bias", "embed_positions.weight",
               "token_embed_positions.weight", "cond_embed.weight"):
        obj = dec
        parts = nm.split(".")
        for p_ in parts[:-1]:
            obj = getattr(obj, p_)
        getattr(obj, parts[-1]).data.copy_(torch.from_numpy(sd[DEC + nm]))
    dec.quantize_codebooks = torch.nn.Parameter(torch.from_numpy(sd[DEC + "quantize_codebooks"]))
    oc = OPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, ffn_dim=cfg.ffn,
                   num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_positions,
                   do_layer_norm_before=False, word_embed_proj_dim=cfg.hidden, activation_function="relu")
    oc._attn_implementation = "eager"
    layers = []
    for i in range(cfg.layers):
        L = OPTDecoderLayer(oc, layer_idx=i).eval()
        sub = {k[len(DEC + f"layers.{i}."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(DEC + f"layers.{i}.")}
        L.load_state_dict(sub, strict=True)
        layers.append(_Layer439(L))
    dec.layers = torch.nn.ModuleList(layers)
    return dec, torch.from_numpy(sd["transformer.lm_head.weight"])


def golden_anchor(out, cfg: MAConfig, sd, lat: torch.Tensor, prefix: torch.Tensor, steps: int = 64):
    """Engine-independent anchor at the 350M shape: the REFERENCE modules' own fp32 numbers for
    (a) a `steps`-step greedy decode of pc_examples/mouse.npy through ShapeOPTDecoder.forward (shape_opt.py:248-438) driven like
        generate() drives it -- per step the chosen token, the top-16 logits, the top-1/top-2 margin and the logits of every 64th
        vocabulary column (eos suppressed, as the throughput configs run);
    (b) the detokenizer's coordinate logits (NoiseResistantDecoder.forward, meshanything.py:50-80; recorded by a forward hook on
        `to_coor_logits`, the reference code path itself is untouched): argmax bin and top-1/top-2 margin of all 800 x 9 coordinates.
    The GPU tests hold the bf16 engine's logits (ma_engine_read_logits) and bins against THESE numbers, not against the oracle."""
    from MeshAnything.models.meshanything import NoiseResistantDecoder, MeshAnything as RefMeshAnything
    dec, lm_head = build_ref_decoder(cfg, sd)
    B, T = 1, cfg.cond_length
    cols = np.arange(0, cfg.vocab, 64)
    toks, top_i, top_v, margin, lcols = [], [], [], [], []

    def record(h):
        lg = (h @ lm_head.T)[0].clone()
        lg[1] = float("-inf")                                  # suppress_eos
        tv, ti = torch.topk(lg, 16)
        toks.append(int(ti[0])); top_i.append(ti.numpy()); top_v.append(tv.numpy()); margin.append(float(tv[0] - tv[1]))
        lcols.append(lg[cols].numpy())
        return int(ti[0])
    with torch.no_grad():
        o = dec(inputs_embeds=prefix, attention_mask=torch.ones(B, T, dtype=torch.long), use_cache=True, return_dict=True)
        pkv = o.past_key_values
        tok = record(o.last_hidden_state[:, -1])
        for t in range(1, steps + 1):
            o = dec(input_ids=torch.tensor([[tok]]), past_key_values=pkv, attention_mask=torch.ones(B, T + t, dtype=torch.long),
                    use_cache=True, return_dict=True)
            pkv = o.past_key_values
            tok = record(o.last_hidden_state[:, -1])
    out["anchor_tokens"] = np.array(toks, dtype=np.int64)          # steps + 1 tokens: from the prefill, then one per step
    out["anchor_top_idx"] = np.stack(top_i).astype(np.int32)
    out["anchor_top_val"] = np.stack(top_v).astype(np.float32)
    out["anchor_margin"] = np.array(margin, dtype=np.float32)
    out["anchor_cols"] = cols.astype(np.int32)
    out["anchor_logits_cols"] = np.stack(lcols).astype(np.float32)
    print(f"anchor decode: {steps + 1} tokens, margin min {min(margin):.4f} median {float(np.median(margin)):.4f}")
    # (b) detokenizer logits of the golden ids (full.npz full_detok_ids), through the reference's own forward
    args = types.SimpleNamespace(codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim)
    tok_m = NoiseResistantDecoder(args)
    sub = {k[len(TOK):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(TOK)}
    tok_m.load_state_dict(sub, strict=True)
    tok_m.eval()
    ids = torch.from_numpy(np.load(os.path.join(HERE, "full.npz"))["full_detok_ids"])
    ns = types.SimpleNamespace(num_quantizers=3)
    ns.transformer = types.SimpleNamespace(model=types.SimpleNamespace(decoder=types.SimpleNamespace(
        quantize_codebooks=torch.from_numpy(sd[DEC + "quantize_codebooks"]))))
    seen = {}
    hook = tok_m.to_coor_logits.register_forward_hook(lambda mod, inp, res: seen.__setitem__("logits", res.detach().clone()))
    with torch.no_grad():
        codes = RefMeshAnything.get_codes(ns, ids)
        coords = tok_m(ids, codes, point_feature=lat)
    hook.remove()
    lg = seen["logits"][0]                                         # (nf, 9, 128)
    tv, ti = torch.topk(lg, 2, dim=-1)
    out["anchor_detok_bins"] = ti[..., 0].numpy().astype(np.int16)
    out["anchor_detok_margin"] = (tv[..., 0] - tv[..., 1]).numpy().astype(np.float32)
    out["anchor_detok_valid"] = (~torch.isnan(coords[0, :, 0, 0])).numpy()
    assert np.array_equal(np.nan_to_num(coords.numpy(), nan=9.0), np.nan_to_num(np.load(os.path.join(HERE, "full.npz"))["full_detok_coords"], nan=9.0))
    m = out["anchor_detok_margin"][out["anchor_detok_valid"]]
    print(f"anchor detok: {int(out['anchor_detok_valid'].sum())} valid faces, margin quantiles 1% {np.quantile(m, 0.01):.4f} 10% {np.quantile(m, 0.1):.4f} 50% {np.quantile(m, 0.5):.4f}")


def golden_anchor_diverse(out, cfg: MAConfig, sd, tag: str, mouse: np.ndarray, mode: str, steps: int = 256):
    """Reference-anchored 350M numbers on a NON-DEGENERATE token stream (the default synthetic checkpoint has a
    fixed point, so `full_anchor.npz` walks a constant stream; an HF-style N(0, 0.02) one cycles through 3 ids -- checkpoint.py,
    init="diverse", says why).  Two ways to a diverse stream, both driven by the REFERENCE's modules on pc_examples/mouse.npy:
      mode "greedy"  (weights init="diverse"): `steps` + 1 greedy tokens through ShapeOPTDecoder.forward, eos suppressed;
      mode "sampled" (weights init="hf", what the reference's constructors leave): the tokens are DRAWN from the reference's own
                     distribution -- transformers' TopKLogitsWarper(50) -> TopPLogitsWarper(0.95), the chain generate(do_sample=True)
                     builds (meshanything.py:153-162) -- by inverse CDF (descending probability, ties by ascending id, fp32 running
                     sum) from seeded uniforms that are stored with the stream.
    Asserted: >= 32 distinct ids.  Recorded per step, fp32: the token fed next, the argmax, the top-16 logits, the top-1/top-2 margin,
    every 64th column; the perceiver's latents / the prefix (rows + 8 columns) on these weights; the detokenizer's bins and margins; and
      * the SAME token path under `torch.autocast("cpu", dtype=torch.float16)` -- the reference's own precision policy
        (main.py:114-118,149: Accelerator(mixed_precision="fp16") + accelerator.autocast()): fp16 Linear inputs / weights / outputs and an
        fp16 KV cache, fp32 LayerNorm and residual stream.  (Attention is transformers' eager path here, which rounds the scores and
        probabilities to fp16; flash-attn keeps them fp32 inside the kernel -- the fixture is the reference's precision CLASS, not
        its flash kernel.)  Stored: the logits at the fp32 top-16 indices and columns, the fp16 argmax and margin per step."""
    assert mode in ("greedy", "sampled")
    from transformers.generation.logits_process import TopKLogitsWarper, TopPLogitsWarper
    from MeshAnything.models.meshanything import NoiseResistantDecoder, MeshAnything as RefMeshAnything
    scratch = {}
    rows = [0, 1, 2, 3, 100, 255, 256]
    lat, prefix = golden_encoder(scratch, cfg, sd, "e", mouse, rows=rows)
    for k in ("rows", "latents_rows", "latents_cols8", "prefix_rows", "prefix_cols8", "stats"):
        out[f"{tag}_{k}"] = scratch[f"e_{k}"]
    dec, lm_head = build_ref_decoder(cfg, sd)
    B, T = 1, cfg.cond_length
    cols = np.arange(0, cfg.vocab, 64)
    toks, top_i, top_v, margin, lcols = [], [], [], [], []

    uni = torch.rand(steps + 1, generator=torch.Generator().manual_seed(2024)).numpy().astype(np.float32)
    warp_k, warp_p = TopKLogitsWarper(top_k=50), TopPLogitsWarper(top_p=0.95)
    kept_n = []

    def record(h):
        lg = (h.float() @ lm_head.T)[0].clone()
        lg[1] = float("-inf")
        tv, ti = torch.topk(lg, 16)
        top_i.append(ti.numpy()); top_v.append(tv.numpy()); margin.append(float(tv[0] - tv[1]))
        lcols.append(lg[cols].numpy())
        tok = int(ti[0])
        if mode == "sampled":
            sc = warp_p(None, warp_k(None, lg[None].clone()))
            probs = torch.softmax(sc, dim=-1)[0]
            kept = torch.nonzero(probs > 0).flatten().tolist()
            kept.sort(key=lambda i: (-float(lg[i]), i))
            kept_n.append(len(kept))
            acc, u, tok = torch.zeros((), dtype=torch.float32), float(uni[len(toks)]), kept[-1]
            for i in kept:
                acc = acc + probs[i]
                if float(acc) > u:
                    tok = i
                    break
        toks.append(tok)
        return tok
    with torch.no_grad():
        o = dec(inputs_embeds=prefix, attention_mask=torch.ones(B, T, dtype=torch.long), use_cache=True, return_dict=True)
        pkv = o.past_key_values
        tok = record(o.last_hidden_state[:, -1])
        for t in range(1, steps + 1):
            o = dec(input_ids=torch.tensor([[tok]]), past_key_values=pkv, attention_mask=torch.ones(B, T + t, dtype=torch.long),
                    use_cache=True, return_dict=True)
            pkv = o.past_key_values
            tok = record(o.last_hidden_state[:, -1])
    distinct = len(set(toks))
    assert distinct >= 32, f"the {tag} stream has only {distinct} distinct tokens: not a diverse anchor"
    out[f"{tag}_mode"] = np.array([0 if mode == "greedy" else 1])
    out[f"{tag}_uniforms"] = uni
    if mode == "sampled":
        out[f"{tag}_kept"] = np.array(kept_n, dtype=np.int32)
    out[f"{tag}_tokens"] = np.array(toks, dtype=np.int64)
    out[f"{tag}_top_idx"] = np.stack(top_i).astype(np.int32)
    out[f"{tag}_top_val"] = np.stack(top_v).astype(np.float32)
    out[f"{tag}_margin"] = np.array(margin, dtype=np.float32)
    out[f"{tag}_cols"] = cols.astype(np.int32)
    out[f"{tag}_logits_cols"] = np.stack(lcols).astype(np.float32)
    print(f"anchor[{tag}] fp32: {steps + 1} tokens, {distinct} distinct, margin min {min(margin):.5f} 10% {np.quantile(margin, 0.1):.4f} median {float(np.median(margin)):.4f}")
    # ---- the same path under fp16 autocast (teacher-forced on the fp32 tokens) ----
    h_top, h_cols, h_arg, h_margin = [], [], [], []
    lm16 = lm_head.half()

    def record16(h, j):
        # lm_head is an nn.Linear in the reference (shape_opt.py:24,155): autocast runs it in fp16 as well
        lg = torch.nn.functional.linear(h, lm16)[0].float()
        lg[1] = float("-inf")
        tv, ti = torch.topk(lg, 2)
        h_arg.append(int(ti[0])); h_margin.append(float(tv[0] - tv[1]))
        h_top.append(lg[torch.from_numpy(top_i[j]).long()].numpy()); h_cols.append(lg[cols].numpy())
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
        o = dec(inputs_embeds=prefix, attention_mask=torch.ones(B, T, dtype=torch.long), use_cache=True, return_dict=True)
        pkv = o.past_key_values
        assert pkv[0][0].dtype == torch.float16                    # the KV cache of the reference's policy
        record16(o.last_hidden_state[:, -1].half(), 0)
        for t in range(1, steps + 1):
            o = dec(input_ids=torch.tensor([[toks[t - 1]]]), past_key_values=pkv, attention_mask=torch.ones(B, T + t, dtype=torch.long),
                    use_cache=True, return_dict=True)
            pkv = o.past_key_values
            record16(o.last_hidden_state[:, -1].half(), t)
    out[f"{tag}_f16_top_val"] = np.stack(h_top).astype(np.float32)
    out[f"{tag}_f16_logits_cols"] = np.stack(h_cols).astype(np.float32)
    out[f"{tag}_f16_argmax"] = np.array(h_arg, dtype=np.int64)
    out[f"{tag}_f16_margin"] = np.array(h_margin, dtype=np.float32)
    err16 = float(np.abs(out[f"{tag}_f16_top_val"] - out[f"{tag}_top_val"]).max())
    agree = float(np.mean(np.array(h_arg) == np.stack(top_i)[:, 0]))
    print(f"anchor[{tag}] fp16 autocast along the fp32 path: max |logit - fp32| over the top-16 {err16:.5f}, argmax agreement {agree * 100:.2f} %")
    # ---- detokenizer on these weights ----
    args = types.SimpleNamespace(codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim)
    tok_m = NoiseResistantDecoder(args)
    sub = {k[len(TOK):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(TOK)}
    tok_m.load_state_dict(sub, strict=True)
    tok_m.eval()
    ids = torch.from_numpy(np.load(os.path.join(HERE, "full.npz"))["full_detok_ids"])
    ns = types.SimpleNamespace(num_quantizers=3)
    ns.transformer = types.SimpleNamespace(model=types.SimpleNamespace(decoder=types.SimpleNamespace(
        quantize_codebooks=torch.from_numpy(sd[DEC + "quantize_codebooks"]))))
    seen = {}
    hook = tok_m.to_coor_logits.register_forward_hook(lambda mod, inp, res: seen.__setitem__("logits", res.detach().clone()))
    with torch.no_grad():
        codes = RefMeshAnything.get_codes(ns, ids)
        coords = tok_m(ids, codes, point_feature=lat)
    hook.remove()
    lg = seen["logits"][0]
    tv, ti = torch.topk(lg, 2, dim=-1)
    out[f"{tag}_detok_bins"] = ti[..., 0].numpy().astype(np.int16)
    out[f"{tag}_detok_margin"] = (tv[..., 0] - tv[..., 1]).numpy().astype(np.float32)
    out[f"{tag}_detok_valid"] = (~torch.isnan(coords[0, :, 0, 0])).numpy()
    m = out[f"{tag}_detok_margin"][out[f"{tag}_detok_valid"]]
    print(f"anchor[{tag}] detok: {int(out[f'{tag}_detok_valid'].sum())} valid faces, margin quantiles 1% {np.quantile(m, 0.01):.4f} 50% {np.quantile(m, 0.5):.4f}")


def golden_anchor_long(cfg: MAConfig, sd, mouse: np.ndarray, steps: int, path: str, every: int = 600):
    """FULL-LENGTH reference anchor: the reference's own ShapeOPTDecoder.forward (shape_opt.py:248-438)
    stepped `steps` times on the `dva` weights exactly as generate() drives it (meshanything.py:140-151: one token per call, tuple
    KV cache grown by torch.cat, attention mask of T + t ones), greedy, eos suppressed -- position rows up to T + steps + 1 of
    `embed_positions` (shape_opt.py:359), every face slot of `token_embed_positions` hundreds of times, context lengths to
    T + steps.  Because the weights of the 800-face and the 1600-face configuration are the same tensors and generate() stops on
    max_new_tokens only, ONE decode of 14 402 steps is the reference stream of BASELINE configs[1] (its first 7 202 tokens) and of
    configs[4] (all of it).  Recorded for EVERY step, fp32: the greedy token, the top-8 logits (values + ids), the top-1/top-2
    margin; at `dense` steps (64 spread over the range + the last 16 of either configuration) the top-16 and every 64th column.
    Partial results are written every `every` steps (the run takes about two hours on 8 cores)."""
    import time
    scratch = {}
    lat, prefix = golden_encoder(scratch, cfg, sd, "e", mouse, rows=[0])
    dec, lm_head = build_ref_decoder(cfg, sd)
    B, T = 1, cfg.cond_length
    cols = np.arange(0, cfg.vocab, 64)
    n = steps + 1
    dense = sorted(set(np.linspace(0, steps, 64).astype(int).tolist()) | set(range(7202 - 16, 7202)) | set(range(steps - 15, steps + 1)))
    dense = [d for d in dense if 0 <= d <= steps]
    dpos = {d: i for i, d in enumerate(dense)}
    toks = np.zeros(n, np.int16); top_i = np.zeros((n, 8), np.int16); top_v = np.zeros((n, 8), np.float32); margin = np.zeros(n, np.float32)
    d_i = np.zeros((len(dense), 16), np.int16); d_v = np.zeros((len(dense), 16), np.float32); d_c = np.zeros((len(dense), len(cols)), np.float32)

    def record(h, j):
        lg = (h.float() @ lm_head.T)[0].clone()
        lg[1] = float("-inf")                                  # suppress_eos (as the throughput configs run)
        tv, ti = torch.topk(lg, 16)
        toks[j] = int(ti[0]); top_i[j] = ti[:8].numpy(); top_v[j] = tv[:8].numpy(); margin[j] = float(tv[0] - tv[1])
        if j in dpos:
            d_i[dpos[j]] = ti.numpy(); d_v[dpos[j]] = tv.numpy(); d_c[dpos[j]] = lg[cols].numpy()
        return int(ti[0])

    def save(upto):
        nd = sum(1 for d in dense if d < upto)
        np.savez_compressed(path + ".tmp.npz", long_tokens=toks[:upto], long_top_idx=top_i[:upto], long_top_val=top_v[:upto], long_margin=margin[:upto],
                            long_dense_steps=np.array(dense[:nd], np.int32), long_dense_top_idx=d_i[:nd], long_dense_top_val=d_v[:nd],
                            long_dense_cols=cols.astype(np.int32), long_dense_logits_cols=d_c[:nd],
                            long_prefix_cols8=prefix[0, :, :8].numpy(), long_complete=np.array([int(upto == n)]))
        os.replace(path + ".tmp.npz", path)
    t0 = time.time()
    with torch.no_grad():
        o = dec(inputs_embeds=prefix, attention_mask=torch.ones(B, T, dtype=torch.long), use_cache=True, return_dict=True)
        pkv = o.past_key_values
        tok = record(o.last_hidden_state[:, -1], 0)
        for t in range(1, steps + 1):
            o = dec(input_ids=torch.tensor([[tok]]), past_key_values=pkv, attention_mask=torch.ones(B, T + t, dtype=torch.long),
                    use_cache=True, return_dict=True)
            pkv = o.past_key_values
            tok = record(o.last_hidden_state[:, -1], t)
            if t % every == 0 or t == 7201:
                save(t + 1)
                print(f"anchor[long] step {t}/{steps}: {time.time() - t0:.0f} s, {len(set(toks[:t + 1].tolist()))} distinct, margin min {margin[:t + 1].min():.5f}", flush=True)
    save(n)
    assert len(set(toks.tolist())) >= min(256, n // 8)
    print(f"anchor[long]: {n} tokens, {len(set(toks.tolist()))} distinct, margin min {margin.min():.5f} 1% {np.quantile(margin, 0.01):.5f} median {np.median(margin):.4f}")


def golden_shapeopt_generate(out, cfg: MAConfig, sd):
    """THE OUTERMOST COMPOSITION: the reference's own `ShapeOPT` CausalLM wrapper
    (shape_opt.py:18-178) -> ShapeOPTModel -> ShapeOPTDecoder.forward, driven by the container's `GenerationMixin.generate` with the
    arguments of meshanything.py:143-151 (inputs_embeds, max_new_tokens, num_beams=1, bos/eos/pad ids).  Three shims bridge
    transformers 4.39.3 -> 5.x, none touches the model's arithmetic or the loop's decisions:
      1. `ShapeOPT.tie_weights` is called with keyword arguments by 5.x's post_init; the reference's override takes none -> the
         keywords are dropped;
      2. 5.x would hand the model a `DynamicCache`; the reference indexes a tuple cache (`past_key_values[0][0].shape[2]`,
         shape_opt.py:342) -> the model declares no Cache-class support, so generate() passes None first and then whatever the
         model returned, as 4.39.3 did;
      3. 4.39.3's `OPTForCausalLM.prepare_inputs_for_generation` (inputs_embeds on the first call only, then the last token; the
         method the reference inherited) is restated, because 5.x's generic one slices by `cache_position`.
    The decoder layers are the container's real OPTDecoderLayer behind the 4.39.3 call signature (`_Layer439`), attention eager."""
    import re
    from MeshAnything.models.shape_opt import ShapeOPT, ShapeOPTConfig
    from transformers import OPTConfig
    from transformers.models.opt.modeling_opt import OPTDecoderLayer
    c = ShapeOPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, ffn_dim=cfg.ffn,
                       num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_positions,
                       do_layer_norm_before=False, word_embed_proj_dim=cfg.hidden, activation_function="relu",
                       bos_token_id=0, eos_token_id=1, pad_token_id=2)
    c.quantize_codebook_dim = cfg.codebook_dim
    c.face_per_token = 9
    c.cond_length = cfg.cond_length
    c._attn_implementation = "eager"
    orig_tie = ShapeOPT.tie_weights
    ShapeOPT.tie_weights = lambda self, *a, **k: orig_tie(self)                                   # shim 1
    ShapeOPT._supports_default_dynamic_cache = classmethod(lambda cls: False)                     # shim 2

    def prepare(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kwargs):   # shim 3 ([3p] 4.39.3 modeling_opt.py)
        if past_key_values is not None:
            past_length = past_key_values[0][0].shape[2]
            remove = past_length if input_ids.shape[1] > past_length else input_ids.shape[1] - 1
            input_ids = input_ids[:, remove:]
        mi = {"inputs_embeds": inputs_embeds} if (inputs_embeds is not None and past_key_values is None) else {"input_ids": input_ids}
        mi.update({"past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"), "attention_mask": attention_mask})
        return mi
    ShapeOPT.prepare_inputs_for_generation = prepare

    def build(sd_):
        m = ShapeOPT(c).eval()
        dec = m.model.decoder
        dec._use_flash_attention_2 = True
        oc = OPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, ffn_dim=cfg.ffn,
                       num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_positions,
                       do_layer_norm_before=False, word_embed_proj_dim=cfg.hidden, activation_function="relu")
        oc._attn_implementation = "eager"
        dec.layers = torch.nn.ModuleList([_Layer439(OPTDecoderLayer(oc, layer_idx=i).eval()) for i in range(cfg.layers)])
        dec.quantize_codebooks = torch.nn.Parameter(torch.zeros(1, cfg.codebook_size, cfg.codebook_dim))      # meshanything.py:118
        # the checkpoint's `transformer.*` keys, loaded strictly (main.py:104); the adapter nests each real layer under `.layer`
        sub = {re.sub(r"(model\.decoder\.layers\.\d+)\.", r"\1.layer.", k[len("transformer."):]): torch.from_numpy(v) for k, v in sd_.items() if k.startswith("transformer.")}
        res = m.load_state_dict(sub, strict=True)
        return m
    g = torch.Generator().manual_seed(6)
    B = 4
    prefix = torch.randn(B, cfg.cond_length, cfg.hidden, generator=g) * 0.7
    m = build(sd)
    with torch.no_grad():
        base = m.generate(inputs_embeds=prefix, max_new_tokens=cfg.max_new_tokens, num_beams=1, bos_token_id=0, eos_token_id=1, pad_token_id=2)
        short = m.generate(inputs_embeds=prefix, max_new_tokens=11, num_beams=1, bos_token_id=0, eos_token_id=1, pad_token_id=2)
    out["gen_prefix"] = prefix.numpy()
    out["gen_tokens"] = base.numpy()
    out["gen_tokens_max11"] = short.numpy()
    # eos firing naturally at different steps per row: swap the lm_head rows of eos and of a token the rows emit at different positions
    pick = None
    for tok in sorted(set(base.flatten().tolist()) - {0, 1, 2}):
        first = [(base[b] == tok).nonzero()[0].item() if (base[b] == tok).any() else None for b in range(B)]
        if len(set(first)) >= 3:
            pick = tok
            break
    assert pick is not None, "no token found that finishes the rows at different steps"
    sd2 = dict(sd)
    w = sd["transformer.lm_head.weight"].copy()
    w[[1, pick]] = w[[pick, 1]]
    sd2["transformer.lm_head.weight"] = w
    m2 = build(sd2)
    with torch.no_grad():
        eos = m2.generate(inputs_embeds=prefix, max_new_tokens=cfg.max_new_tokens, num_beams=1, bos_token_id=0, eos_token_id=1, pad_token_id=2)
    out["gen_eos_swap_token"] = np.array([pick])
    out["gen_tokens_eos"] = eos.numpy()
    ShapeOPT.tie_weights = orig_tie
    print(f"ShapeOPT under GenerationMixin.generate: {tuple(base.shape)} tokens; eos variant (rows 1 <-> {pick} of lm_head) {tuple(eos.shape)}, "
          f"row lengths {[int((eos[b] == 1).nonzero()[0]) + 1 if (eos[b] == 1).any() else eos.shape[1] for b in range(B)]}")


def golden_detok(out, cfg: MAConfig, sd, tag: str, lat: torch.Tensor, seed: int):
    from MeshAnything.models.meshanything import NoiseResistantDecoder, MeshAnything as RefMeshAnything, undiscretize
    import transformers
    if cfg.tok_width != 768:
        _orig = transformers.AutoConfig.from_pretrained

        def _fp(name, *a, **k):
            return transformers.BertConfig(hidden_size=cfg.tok_width, num_attention_heads=cfg.tok_heads,
                                           intermediate_size=cfg.tok_ffn)
        transformers.AutoConfig.from_pretrained = staticmethod(_fp)
    args = types.SimpleNamespace(codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim)
    tok = NoiseResistantDecoder(args)
    if cfg.tok_width != 768:
        transformers.AutoConfig.from_pretrained = staticmethod(_orig)
        # the reference hard-codes these (meshanything.py:27,31-35); rebuild them at the test shape
        tok.pos_embedding = torch.nn.Embedding(cfg.tok_max_pos, cfg.tok_width)
        tok.cond_length = cfg.cond_length
        tok.cond_dim = cfg.enc_width
        tok.point_pe = torch.nn.Embedding(cfg.cond_length, cfg.tok_width)
        tok.cond_proj = torch.nn.Linear(cfg.enc_width, cfg.tok_width)
        tok.cond_head_proj = torch.nn.Linear(cfg.enc_width, cfg.tok_width)
    assert len(tok.decoder.layer) == 6
    tok.decoder.layer = tok.decoder.layer[:cfg.tok_layers]
    sub = {k[len(TOK):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(TOK)}
    tok.load_state_dict(sub, strict=True)
    tok.eval()
    rng = np.random.default_rng(seed)
    nf = cfg.n_max_faces
    ids = rng.integers(0, cfg.codebook_size, size=(1, nf * 9)).astype(np.int64)
    n_valid = max(2, nf * 2 // 3)
    ids[0, n_valid * 9:] = -1                     # tail faces padded (stream ended)
    ids[0, 9 * 1 + 4] = -1                        # a special token mid-sequence kills just that face
    ids_t = torch.from_numpy(ids)
    ns = types.SimpleNamespace(num_quantizers=3)
    ns.transformer = types.SimpleNamespace(model=types.SimpleNamespace(decoder=types.SimpleNamespace(
        quantize_codebooks=torch.from_numpy(sd[DEC + "quantize_codebooks"]))))
    with torch.no_grad():
        codes = RefMeshAnything.get_codes(ns, ids_t)                  # meshanything.py:178-212, unbound
        coords = tok(ids_t, codes, point_feature=lat)                 # meshanything.py:50-80
        # logits for margin information (same modules, same order as forward 50-69)
        pf = tok.process_point_feature(lat)
    out[f"{tag}_detok_ids"] = ids
    out[f"{tag}_detok_codes_rows"] = codes[0, [0, 1, 5, 3 * n_valid - 1, 3 * nf - 1]].numpy()
    out[f"{tag}_detok_codes_stats"] = np.array([codes.double().sum(), codes.double().abs().sum()])
    out[f"{tag}_detok_coords"] = coords.numpy()
    out[f"{tag}_detok_pf_rows"] = pf[0, [0, 1, cfg.cond_length - 1]].numpy()
    out["undiscretize"] = undiscretize(torch.arange(128), low=-0.5, high=0.5, num_discrete=128).numpy()
    print(f"detok[{tag}]: valid faces", int((~torch.isnan(coords[0, :, 0, 0])).sum()), "of", nf)


def golden_warpers(out):
    """[3p] TopKLogitsWarper(50) -> TopPLogitsWarper(0.95) -> softmax, the chain `generate(do_sample=True, top_k=50,
    top_p=0.95)` builds (meshanything.py:153-162)."""
    from transformers.generation.logits_process import TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(5)
    rows = []
    for V, scale in ((8195, 1.0), (8195, 4.0), (64, 1.0), (64, 0.05), (8195, 0.2)):
        logits = torch.randn(1, V, generator=g) * scale
        s = TopKLogitsWarper(top_k=50)(None, logits.clone())
        s = TopPLogitsWarper(top_p=0.95)(None, s)
        probs = torch.softmax(s, dim=-1)[0]
        rows.append((logits[0].numpy(), probs.numpy()))
    for i, (lg, pr) in enumerate(rows):
        out[f"warp_logits_{i}"] = lg
        out[f"warp_probs_{i}"] = pr
    out["warp_n"] = np.array([len(rows)])
    print("warpers: kept counts", [int((pr > 0).sum()) for _, pr in rows])


def main():
    _stub_modules()
    torch.manual_seed(0)
    torch.set_num_threads(int(os.environ.get("MA_GOLDEN_THREADS", 8)))
    if "--only-shapeopt" in sys.argv:                     # the other fixtures are committed and unchanged
        tiny = MAConfig.tiny()
        g = {}
        golden_shapeopt_forward(g, tiny, synthetic_state_dict(tiny, include_unused=True))
        np.savez_compressed(os.path.join(HERE, "shapeopt_forward.npz"), **g)
        print("shapeopt_forward.npz", os.path.getsize(os.path.join(HERE, "shapeopt_forward.npz")) // 1024, "KiB")
        return
    if "--only-generate" in sys.argv:                     # the outermost pin: ShapeOPT under GenerationMixin.generate (tiny shape)
        tiny = MAConfig.tiny()
        g = {}
        golden_shapeopt_generate(g, tiny, synthetic_state_dict(tiny, include_unused=True))
        np.savez_compressed(os.path.join(HERE, "shapeopt_generate.npz"), **g)
        print("shapeopt_generate.npz", os.path.getsize(os.path.join(HERE, "shapeopt_generate.npz")) // 1024, "KiB")
        return
    if "--only-anchor-long" in sys.argv:                  # one 14 402-step reference decode (dva weights): configs[1] and configs[4] at full length
        full = MAConfig.full(n_max_faces=1600)
        mouse = np.load(os.path.join(HERE, "dataset.npz"))["mouse_norm"]
        steps = int(os.environ.get("MA_GOLDEN_LONG_STEPS", full.max_new_tokens))      # generated tokens = steps + 1 (the prefill's pick first)
        tag, seed, init, mode = ANCHOR_HF_SETS[0]
        golden_anchor_long(full, synthetic_state_dict(full, seed=seed, include_unused=True, init=init), mouse, steps - 1,
                           os.path.join(HERE, "full_anchor_long.npz"))
        print("full_anchor_long.npz", os.path.getsize(os.path.join(HERE, "full_anchor_long.npz")) // 1024, "KiB")
        return
    if "--only-anchor-hf" in sys.argv:                    # 350M-shape reference numbers on diverse streams (two HF-style weight sets)
        full = MAConfig.full()
        mouse = np.load(os.path.join(HERE, "dataset.npz"))["mouse_norm"]
        g = {}
        for tag, seed, init, mode in ANCHOR_HF_SETS:
            golden_anchor_diverse(g, full, synthetic_state_dict(full, seed=seed, include_unused=True, init=init), tag, mouse, mode)
        np.savez_compressed(os.path.join(HERE, "full_anchor_hf.npz"), **g)
        print("full_anchor_hf.npz", os.path.getsize(os.path.join(HERE, "full_anchor_hf.npz")) // 1024, "KiB")
        return
    if "--only-anchor" in sys.argv:                       # 350M-shape reference numbers for the bf16 engine's logits / bins
        full = MAConfig.full()
        sd_f = synthetic_state_dict(full, include_unused=True)
        mouse = np.load(os.path.join(HERE, "dataset.npz"))["mouse_norm"]
        scratch = {}
        lat_f, prefix_f = golden_encoder(scratch, full, sd_f, "full", mouse, rows=[0])
        g = {}
        golden_anchor(g, full, sd_f, lat_f, prefix_f)
        np.savez_compressed(os.path.join(HERE, "full_anchor.npz"), **g)
        print("full_anchor.npz", os.path.getsize(os.path.join(HERE, "full_anchor.npz")) // 1024, "KiB")
        return
    g = {}
    golden_dataset(g)
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **g)

    tiny = MAConfig.tiny()
    sd_t = synthetic_state_dict(tiny, include_unused=True)
    g = {}
    cloud = synth_cloud(1, tiny.n_points)
    from oracle.meshanything_oracle import normalize_pc
    pc_t = normalize_pc(cloud)
    lat_t, _ = golden_encoder(g, tiny, sd_t, "tiny", pc_t, rows=list(range(tiny.cond_length)))
    golden_decoder_embed(g, tiny, sd_t)
    golden_opt_layers(g, tiny, sd_t)
    golden_detok(g, tiny, sd_t, "tiny", lat_t, seed=21)
    golden_warpers(g)
    np.savez_compressed(os.path.join(HERE, "tiny.npz"), **g)
    g = {}
    golden_shapeopt_forward(g, tiny, sd_t)
    np.savez_compressed(os.path.join(HERE, "shapeopt_forward.npz"), **g)

    full = MAConfig.full()
    sd_f = synthetic_state_dict(full, include_unused=True)
    g = {}
    mouse = np.load(os.path.join(HERE, "dataset.npz"))["mouse_norm"]
    lat_f, prefix_f = golden_encoder(g, full, sd_f, "full", mouse, rows=[0, 1, 2, 3, 100, 255, 256])
    del g["full_input"]                      # = dataset.npz mouse_norm
    golden_detok(g, full, sd_f, "full", lat_f, seed=22)
    np.savez_compressed(os.path.join(HERE, "full.npz"), **g)
    g = {}
    golden_shapeopt_generate(g, tiny, sd_t)
    np.savez_compressed(os.path.join(HERE, "shapeopt_generate.npz"), **g)
    g = {}
    golden_anchor(g, full, sd_f, lat_f, prefix_f)
    np.savez_compressed(os.path.join(HERE, "full_anchor.npz"), **g)
    g = {}
    for tag, seed, init, mode in ANCHOR_HF_SETS:
        golden_anchor_diverse(g, full, synthetic_state_dict(full, seed=seed, include_unused=True, init=init), tag, mouse, mode)
    np.savez_compressed(os.path.join(HERE, "full_anchor_hf.npz"), **g)
    for f in ("dataset.npz", "tiny.npz", "full.npz", "shapeopt_generate.npz", "full_anchor.npz", "full_anchor_hf.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
