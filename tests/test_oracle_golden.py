"""Pin the CPU oracle against outputs of the reference's own code (tests/golden/*.npz, written by
tests/golden/make_golden.py from /root/reference + the container's transformers copy)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from meshanything_amd.config import MAConfig
from meshanything_amd.checkpoint import synthetic_state_dict, state_dict_spec, bf16_round
from oracle.meshanything_oracle import Oracle, normalize_pc, undiscretize, bf16r

torch.set_num_threads(max(1, os.cpu_count() or 1))


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


@pytest.fixture(scope="module")
def tiny(golden_dir):
    cfg = MAConfig.tiny()
    return cfg, Oracle(cfg, synthetic_state_dict(cfg), "fp32"), _load(golden_dir, "tiny.npz")


def test_dataset_normalisation_matches_reference(golden_dir):
    g = _load(golden_dir, "dataset.npz")
    np.random.seed(0)
    idx = np.random.choice(g["mouse_raw"].shape[0], 4096, replace=False)      # main.py:25
    assert idx[:8].tolist() == [582, 1961, 1957, 3193, 3818, 2205, 581, 3779]  # SURVEY.md 8c probe
    out = normalize_pc(g["mouse_raw"][idx])
    assert out.dtype == np.float16
    assert hashlib.sha256(out.tobytes()).hexdigest() == "c6598e1b61ee02dd96cd9790c64313bb152c711c4776c462118744fdb06e52f8"
    assert np.array_equal(out.view(np.uint16), g["mouse_norm"].view(np.uint16))
    np.random.seed(3)
    idx = np.random.choice(g["synth_raw"].shape[0], 4096, replace=False)
    assert np.array_equal(normalize_pc(g["synth_raw"][idx]).view(np.uint16), g["synth_norm"].view(np.uint16))


def test_undiscretize_matches_reference(tiny):
    _, _, g = tiny
    got = undiscretize(torch.arange(128), low=-0.5, high=0.5, num_discrete=128).numpy()
    assert np.array_equal(got, g["undiscretize"])
    assert got[0] == -0.5 and got[127] == 0.4921875


def test_fourier_embed_bit_exact(tiny):
    cfg, o, g = tiny
    x = torch.from_numpy(g["tiny_input"].astype(np.float32))
    got = o.fourier_embed(x[:16, :3]).numpy()
    assert np.array_equal(got, g["tiny_fourier16"])


def test_encoder_tiny_matches_reference(tiny):
    cfg, o, g = tiny
    x = torch.from_numpy(g["tiny_input"].astype(np.float32))[None]
    lat = o.encode_latents(x)
    np.testing.assert_allclose(lat[0, g["tiny_rows"]].numpy(), g["tiny_latents_rows"], atol=1e-5, rtol=0)
    shp = o.to_shape_latents(lat[:, 1:])
    np.testing.assert_allclose(shp[0].numpy()[:, :8], g["tiny_shape_cols8"], atol=2e-5, rtol=0)
    prefix = o.process_point_feature(lat)
    np.testing.assert_allclose(prefix[0, g["tiny_rows"]].numpy(), g["tiny_prefix_rows"], atol=2e-5, rtol=0)


def test_decoder_embedding_matches_reference(tiny):
    cfg, o, g = tiny
    e = o.embed_tokens(torch.from_numpy(g["dec_embed_tok"]), torch.from_numpy(g["dec_embed_t"]))
    np.testing.assert_allclose(e.numpy(), g["dec_embed_e"], atol=2e-6, rtol=0)
    T = cfg.cond_length
    pre = torch.from_numpy(np.arange(T * cfg.hidden, dtype=np.float32).reshape(1, T, cfg.hidden) * 1e-3)
    h0 = o.embed_prefix(pre)
    np.testing.assert_allclose(h0[0, [0, 1, T - 1]].numpy(), g["dec_prefill_h0_rows"], atol=1e-6, rtol=0)


def test_face_slot_rows(tiny):
    """SURVEY.md 3.3 probe: slot rows for t=1..23 with a special first token are [0,3,4,...,11,3,4,...]."""
    cfg, o, _ = tiny
    tab = o.sd["transformer.model.decoder.token_embed_positions.weight"]
    ids = torch.tensor([0] + [5] * 22)
    t = torch.arange(1, 24)
    special = ids < 3
    slot = torch.where(special, ids, torch.remainder(t - 2, 9) + 3)
    assert slot.tolist() == [0] + [3 + ((k - 2) % 9) for k in range(2, 24)]
    assert slot.tolist()[:11] == [0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 3]
    # a non-special first token uses python-modulo semantics: (1-2) % 9 + 3 = 11
    assert int(torch.remainder(torch.tensor(1 - 2), 9) + 3) == 11


def test_opt_layers_match_transformers_and_cache_is_consistent(tiny):
    cfg, o, g = tiny
    h = torch.from_numpy(g["opt_h_in"])
    full = o.opt_layers(h, None)
    np.testing.assert_allclose(full.numpy(), g["opt_h_out"], atol=2e-5, rtol=0)
    # prefill + cached single steps == one causal pass (what generate() relies on)
    T = cfg.cond_length
    cache = [None] * cfg.layers
    outs = [o.opt_layers(h[:, :T], cache)]
    for j in range(T, h.shape[1]):
        outs.append(o.opt_layers(h[:, j:j + 1], cache))
    np.testing.assert_allclose(torch.cat(outs, 1).numpy(), full.numpy(), atol=2e-5, rtol=0)


def test_decoder_forward_matches_the_reference_forward(golden_dir):
    """The reference's OWN `ShapeOPTDecoder.forward` (shape_opt.py:248-438), driven like generate() drives it (prefix call, then
    cached single-token calls with the growing attention mask), vs the oracle's embed_prefix / embed_tokens / opt_layers /
    lm_head: hidden state after the prefill and after each of 13 steps, for two rows with specials mid-sequence."""
    cfg = MAConfig.tiny()
    o = Oracle(cfg, synthetic_state_dict(cfg), "fp32")
    g = _load(golden_dir, "shapeopt_forward.npz")
    prefix = torch.from_numpy(g["sopt_prefix"])
    toks = torch.from_numpy(g["sopt_tokens"])
    B, steps = toks.shape
    assert int(g["sopt_cache_len"][0]) == cfg.cond_length + steps
    for b in range(B):
        cache = [None] * cfg.layers
        h = o.opt_layers(o.embed_prefix(prefix[b:b + 1]), cache)
        got = [h[0, -1]]
        for t in range(1, steps + 1):
            e = o.embed_tokens(toks[b, t - 1:t], torch.tensor([t]))
            got.append(o.opt_layers(e[None], cache)[0, -1])
        got = torch.stack(got)
        np.testing.assert_allclose(got.numpy(), g["sopt_hidden"][b], atol=3e-5, rtol=0)
        logits = o.lm_head(got)
        np.testing.assert_allclose(logits[:, :48].numpy(), g["sopt_logits_cols"][b], atol=1e-4, rtol=0)
        assert np.array_equal(logits.argmax(-1).numpy(), g["sopt_logits_argmax"][b])
        assert cache[0][0].shape[1] == cfg.cond_length + steps


def test_detokenizer_tiny_matches_reference(tiny):
    cfg, o, g = tiny
    x = torch.from_numpy(g["tiny_input"].astype(np.float32))[None]
    lat = o.encode_latents(x)
    ids = torch.from_numpy(g["tiny_detok_ids"])
    codes = o.get_codes(ids)
    rows = [0, 1, 5, 3 * max(2, cfg.n_max_faces * 2 // 3) - 1, 3 * cfg.n_max_faces - 1]
    np.testing.assert_allclose(codes[0, rows].numpy(), g["tiny_detok_codes_rows"], atol=1e-6, rtol=0)
    pf = o.detok_point_feature(lat)
    np.testing.assert_allclose(pf[0, [0, 1, cfg.cond_length - 1]].numpy(), g["tiny_detok_pf_rows"], atol=2e-5, rtol=0)
    coords = o.detokenize(ids, codes, lat).numpy()
    ref = g["tiny_detok_coords"]
    assert np.array_equal(np.isnan(coords), np.isnan(ref))
    assert np.array_equal(np.nan_to_num(coords, nan=9.0), np.nan_to_num(ref, nan=9.0))


def test_topk_topp_matches_transformers_warpers(tiny):
    _, o, g = tiny
    for i in range(int(g["warp_n"][0])):
        logits = torch.from_numpy(g[f"warp_logits_{i}"])
        ref = g[f"warp_probs_{i}"]
        kept, probs = Oracle.topk_topp_filter(logits)
        assert sorted(kept.tolist()) == np.nonzero(ref > 0)[0].tolist()
        np.testing.assert_allclose(probs.numpy(), ref[kept.numpy()], atol=1e-6, rtol=0)
        # inverse-CDF draw: u just inside each interval selects that token
        c = np.cumsum(probs.numpy().astype(np.float64))
        assert Oracle.sample_from(kept, probs, 0.0) == int(kept[0])
        mid = 0.5 * (c[0] + c[1]) if len(c) > 1 else 0.5
        assert Oracle.sample_from(kept, probs, float(mid)) == int(kept[min(1, len(c) - 1)])
        assert Oracle.sample_from(kept, probs, 0.9999999) == int(kept[-1])


def test_postprocess_tokens_semantics():
    cfg = MAConfig.tiny()
    o = Oracle(cfg, {}, "fp32")
    L = cfg.max_new_tokens
    res = torch.tensor([[0] + [10, 11, 12, 13, 14, 15, 16, 17, 18] + [1], [0] + [3, 4, 2, 6, 7, 8, 9, 10, 11] + [12]])
    ids = o.postprocess_tokens(res)
    assert ids.shape == (2, L - 2)
    assert ids[0, :9].tolist() == [7, 8, 9, 10, 11, 12, 13, 14, 15]
    assert (ids[0, 9:] == -1).all()                       # eos and the eos padding become -1
    assert ids[1, :10].tolist() == [0, 1, -1, 3, 4, 5, 6, 7, 8, 9]   # a mid-sequence special becomes -1
    codes = torch.zeros(1)  # noqa: F841


def test_bf16_rounding_helpers_agree():
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32) * 3
    x[:4] = [0.0, -0.0, 1.0000001, 65504.0]
    assert np.array_equal(bf16_round(x), bf16r(torch.from_numpy(x)).numpy())


def test_state_dict_spec_totals():
    full = MAConfig.full()
    spec = state_dict_spec(full, include_unused=True)
    n = sum(int(np.prod(s)) for s, _ in spec.values())
    assert n == 595_837_185          # ~596 M parameters
    assert spec["transformer.model.decoder.embed_positions.weight"][0] == (18261, 1024)
    assert spec["tokenizer.to_coor_logits.0.weight"][0] == (1152, 768)
    fused = state_dict_spec(full, include_unused=True, bert_fused=True)
    assert sum(int(np.prod(s)) for s, _ in fused.values()) == n


def test_bert_fused_names_are_accepted():
    cfg = MAConfig.tiny()
    a = Oracle(cfg, synthetic_state_dict(cfg), "fp32")
    sd = synthetic_state_dict(cfg, bert_fused=True)
    # make the fused tensors carry the same numbers as the vanilla ones
    van = synthetic_state_dict(cfg)
    for n in range(cfg.tok_layers):
        p = f"tokenizer.decoder.layer.{n}."
        sd[p + "in_proj_weight"] = np.concatenate([van[p + f"attention.self.{k}.weight"] for k in ("query", "key", "value")])
        sd[p + "in_proj_bias"] = np.concatenate([van[p + f"attention.self.{k}.bias"] for k in ("query", "key", "value")])
        for a_, b_ in (("out_proj_weight", "attention.output.dense.weight"), ("out_proj_bias", "attention.output.dense.bias"),
                       ("linear1_weight", "intermediate.dense.weight"), ("linear1_bias", "intermediate.dense.bias"),
                       ("linear2_weight", "output.dense.weight"), ("linear2_bias", "output.dense.bias"),
                       ("norm1_weight", "attention.output.LayerNorm.weight"), ("norm1_bias", "attention.output.LayerNorm.bias"),
                       ("norm2_weight", "output.LayerNorm.weight"), ("norm2_bias", "output.LayerNorm.bias")):
            sd[p + a_] = van[p + b_]
    b = Oracle(cfg, sd, "fp32")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, cfg.cond_length + 3, cfg.tok_width, generator=g)
    assert torch.equal(a._bert_layer(x, "tokenizer.decoder.layer.0."), b._bert_layer(x, "tokenizer.decoder.layer.0."))


@pytest.mark.slow
def test_encoder_and_detok_full_size_match_reference(golden_dir):
    """350M-shape encoder + detokenizer on pc_examples/mouse.npy vs the reference's own modules."""
    cfg = MAConfig.full()
    g = _load(golden_dir, "full.npz")
    d = _load(golden_dir, "dataset.npz")
    spec = state_dict_spec(cfg)
    need = {k: v for k, v in spec.items() if k.startswith("point_encoder.") or k.startswith("tokenizer.")
            or k.startswith("cond_") or k.endswith("quantize_codebooks")}
    from meshanything_amd.checkpoint import synthetic_tensor
    sd = {k: synthetic_tensor(cfg, k, s, kind) for k, (s, kind) in need.items()}
    o = Oracle(cfg, sd, "fp32")
    x = torch.from_numpy(d["mouse_norm"].astype(np.float32))[None]
    lat = o.encode_latents(x)
    rows = g["full_rows"]
    np.testing.assert_allclose(lat[0, rows].numpy(), g["full_latents_rows"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(lat[0, :, :8].numpy(), g["full_latents_cols8"], atol=1e-5, rtol=0)
    prefix = o.process_point_feature(lat)
    np.testing.assert_allclose(prefix[0, rows].numpy(), g["full_prefix_rows"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(prefix[0, :, :8].numpy(), g["full_prefix_cols8"], atol=5e-5, rtol=0)
    ids = torch.from_numpy(g["full_detok_ids"])
    codes = o.get_codes(ids)
    st = g["full_detok_codes_stats"]
    assert abs(float(codes.double().sum()) - st[0]) < 1e-3 * max(1.0, abs(st[0]))
    coords = o.detokenize(ids, codes, lat).numpy()
    ref = g["full_detok_coords"]
    assert np.array_equal(np.isnan(coords), np.isnan(ref))
    mism = int((np.nan_to_num(coords, nan=9.0) != np.nan_to_num(ref, nan=9.0)).sum())
    assert mism <= 2, f"{mism} coordinate bins differ"     # argmax near-ties between two fp32 summation orders
