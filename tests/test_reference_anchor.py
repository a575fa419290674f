"""The oracle against the two fixtures that pin the OUTERMOST layers of the reference (tests/golden/make_golden.py
--only-generate / --only-anchor, generated in the authoring container from /root/reference):

  shapeopt_generate.npz  the reference's own `ShapeOPT` CausalLM wrapper (shape_opt.py:18-178) driven by HuggingFace
                         `GenerationMixin.generate` with the call of meshanything.py:143-151 -- tokens of a full-length greedy run, a
                         truncated run, and a run where rows hit eos at different steps (tiny shape);
  full_anchor.npz        350M shape, pc_examples/mouse.npy: 65 greedy tokens + their top-16 logits through the reference's own
                         ShapeOPTDecoder.forward, and the detokenizer's coordinate logits (bins + margins) from
                         NoiseResistantDecoder.forward.
The GPU suite (tests/test_gpu_reference_anchor.py) holds the HIP engine against the same files."""
import os

import numpy as np
import pytest
import torch

from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.config import MAConfig
from oracle.meshanything_oracle import Oracle


def test_oracle_generate_equals_shapeopt_under_hf_generate(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "shapeopt_generate.npz")))
    cfg = MAConfig.tiny()
    sd = synthetic_state_dict(cfg)
    o = Oracle(cfg, sd, "fp32")
    prefix = torch.from_numpy(g["gen_prefix"])
    assert torch.equal(o.generate(prefix), torch.from_numpy(g["gen_tokens"]))
    assert torch.equal(o.generate(prefix, max_new_tokens=11), torch.from_numpy(g["gen_tokens_max11"]))
    tok = int(g["gen_eos_swap_token"][0])
    sd2 = dict(sd)
    w = sd["transformer.lm_head.weight"].copy()
    w[[1, tok]] = w[[tok, 1]]
    sd2["transformer.lm_head.weight"] = w
    ref = torch.from_numpy(g["gen_tokens_eos"])
    got = Oracle(cfg, sd2, "fp32").generate(prefix)
    assert torch.equal(got, ref)                                   # eos per row, pad=2 afterwards, width = the slowest row
    lens = [int((ref[b] == 1).nonzero()[0]) + 1 if (ref[b] == 1).any() else ref.shape[1] for b in range(ref.shape[0])]
    assert len(set(lens)) >= 3 and all((ref[b, n:] == 2).all() for b, n in enumerate(lens))


def test_oracle_matches_the_350m_reference_anchor(golden_dir, state_dicts):
    a = dict(np.load(os.path.join(golden_dir, "full_anchor.npz")))
    d = dict(np.load(os.path.join(golden_dir, "dataset.npz")))
    cfg = MAConfig.full()
    sd = state_dicts(cfg)
    o = Oracle(cfg, sd, "fp32")
    x = torch.from_numpy(d["mouse_norm"])[None]
    lat = o.encode_latents(x)
    prefix = o.process_point_feature(lat)
    n = int(os.environ.get("MA_TEST_ANCHOR_STEPS", "24"))
    toks, logits = o.generate(prefix, max_new_tokens=n, suppress_eos=True, return_logits=True)
    assert toks[0].tolist() == a["anchor_tokens"][:n].tolist()
    for j in range(n):                                             # row_logits[j] chose token j
        lg = logits[0][j].clone()
        lg[1] = float("-inf")
        err = float((lg[torch.from_numpy(a["anchor_top_idx"][j]).long()] - torch.from_numpy(a["anchor_top_val"][j])).abs().max())
        err = max(err, float((lg[torch.from_numpy(a["anchor_cols"]).long()] - torch.from_numpy(a["anchor_logits_cols"][j])).abs().max()))
        assert err < 2e-4, (j, err)
    ids = torch.from_numpy(dict(np.load(os.path.join(golden_dir, "full.npz")))["full_detok_ids"])
    _, dl = o.detokenize(ids, o.get_codes(ids), lat, return_logits=True)
    tv, ti = torch.topk(dl[0], 2, dim=-1)
    valid = torch.from_numpy(a["anchor_detok_valid"])
    assert torch.equal(ti[..., 0][valid], torch.from_numpy(a["anchor_detok_bins"]).long()[valid])
    assert float(((tv[..., 0] - tv[..., 1])[valid] - torch.from_numpy(a["anchor_detok_margin"])[valid]).abs().max()) < 2e-4


@pytest.mark.parametrize("tag,init", [("dva", "diverse"), ("hfa", "hf")])
def test_oracle_matches_the_diverse_350m_anchors(tag, init, golden_dir, state_dicts):
    """full_anchor_hf.npz (round 4): 257 steps along a reference-produced path with >= 32 distinct token ids -- greedy on the
    init="diverse" weights, drawn from the reference's own top-k/top-p distribution (transformers' warpers, stored uniforms) on the
    HF-style ones.  The oracle, teacher-forced along the stored tokens, must reproduce the reference's logits at EVERY step, its argmax
    wherever the margin is above rounding, the warpers' draw, the perceiver's activations and the detokenizer's bins."""
    a = dict(np.load(os.path.join(golden_dir, "full_anchor_hf.npz")))
    d = dict(np.load(os.path.join(golden_dir, "dataset.npz")))
    cfg = MAConfig.full()
    o = Oracle(cfg, state_dicts(cfg, init=init), "fp32")
    x = torch.from_numpy(d["mouse_norm"])[None]
    lat = o.encode_latents(x)
    prefix = o.process_point_feature(lat)
    rows = a[f"{tag}_rows"]
    assert float(np.abs(lat[0, rows].numpy() - a[f"{tag}_latents_rows"]).max()) < 1e-5
    assert float(np.abs(prefix[0, rows].numpy() - a[f"{tag}_prefix_rows"]).max()) < 5e-5
    toks = torch.from_numpy(a[f"{tag}_tokens"])
    n = toks.shape[0]
    assert len(set(toks.tolist())) >= 32
    lg = o.teacher_forced_logits(prefix, toks)[:n].clone()
    lg[:, 1] = float("-inf")
    top_i = torch.from_numpy(a[f"{tag}_top_idx"]).long()
    err = max(float((lg.gather(1, top_i) - torch.from_numpy(a[f"{tag}_top_val"])).abs().max()),
              float((lg[:, torch.from_numpy(a[f"{tag}_cols"]).long()] - torch.from_numpy(a[f"{tag}_logits_cols"])).abs().max()))
    assert err < 5e-4, err
    margin = torch.from_numpy(a[f"{tag}_margin"])
    clear = margin > 1e-3
    assert torch.equal(lg.argmax(dim=1)[clear], top_i[:, 0][clear])
    if int(a[f"{tag}_mode"][0]) == 1:              # the oracle's restated warpers + inverse-CDF draw against transformers' own, at the 350M vocabulary
        same = 0
        for j in range(n):
            kept, probs = Oracle.topk_topp_filter(lg[j])
            assert abs(len(kept) - int(a[f"{tag}_kept"][j])) <= 1          # a cumulative mass within rounding of the cut may fall either side
            same += int(Oracle.sample_from(kept, probs, float(a[f"{tag}_uniforms"][j])) == int(toks[j]))
        assert same >= n - 3, same
    else:
        assert torch.equal(lg.argmax(dim=1)[clear], toks[clear])
    ids = torch.from_numpy(dict(np.load(os.path.join(golden_dir, "full.npz")))["full_detok_ids"])
    _, dl = o.detokenize(ids, o.get_codes(ids), lat, return_logits=True)
    tv, ti = torch.topk(dl[0], 2, dim=-1)
    valid = torch.from_numpy(a[f"{tag}_detok_valid"])
    dm = torch.from_numpy(a[f"{tag}_detok_margin"])
    ok = valid[:, None] & (dm > 1e-3)
    assert torch.equal(ti[..., 0][ok], torch.from_numpy(a[f"{tag}_detok_bins"]).long()[ok])


def test_oracle_matches_the_full_length_anchor(golden_dir, state_dicts):
    """full_anchor_long.npz (round 5): ONE greedy decode of the reference's own ShapeOPTDecoder.forward over 14 402 tokens (the stream of BASELINE
    configs[1] = its first 7 202 tokens, and of configs[4]).  The fixture is consistent with the 257-step one (same weights, same cloud: the
    same first tokens and logits), and the oracle -- teacher-forced in one causal pass over the first MA_TEST_LONG_STEPS tokens (CPU: the
    whole stream would be 20 TFLOP; the GPU suite walks all of it, tests/test_gpu_long_context.py) -- reproduces the reference's top-8 logits
    on every step and its token wherever the margin is above rounding."""
    a = dict(np.load(os.path.join(golden_dir, "full_anchor_long.npz")))
    h = dict(np.load(os.path.join(golden_dir, "full_anchor_hf.npz")))
    d = dict(np.load(os.path.join(golden_dir, "dataset.npz")))
    assert int(a["long_complete"][0]) == 1 and len(a["long_tokens"]) == 14402
    toks_all = torch.from_numpy(a["long_tokens"].astype(np.int64))
    assert len(set(toks_all[:7202].tolist())) >= 256 and len(set(toks_all.tolist())) >= 400
    assert int(toks_all.min()) >= 0 and int(toks_all.max()) < 8195 and not bool((toks_all == 1).any())          # eos suppressed
    assert toks_all[:257].tolist() == h["dva_tokens"].tolist()
    assert float(np.abs(a["long_top_val"][:257] - h["dva_top_val"][:, :8]).max()) < 2e-5                          # (two runs of the reference, other thread counts)
    assert np.array_equal(a["long_top_idx"][:, 0].astype(np.int64), a["long_tokens"].astype(np.int64))           # greedy: the token IS the argmax
    ds = a["long_dense_steps"]
    assert len(ds) >= 64 and ds[-1] == 14401 and (ds >= 7000).sum() >= 16 and {7186, 7201}.issubset(set(ds.tolist()))
    n = int(os.environ.get("MA_TEST_LONG_STEPS", "700"))
    cfg = MAConfig.full()
    o = Oracle(cfg, state_dicts(cfg, init="diverse"), "fp32")
    x = torch.from_numpy(d["mouse_norm"])[None]
    prefix = o.process_point_feature(o.encode_latents(x))
    assert float(np.abs(prefix[0, :, :8].numpy() - a["long_prefix_cols8"]).max()) < 5e-5
    lg = o.teacher_forced_logits(prefix, toks_all[:n])[:n].clone()
    lg[:, 1] = float("-inf")
    top_i = torch.from_numpy(a["long_top_idx"][:n].astype(np.int64))
    err = float((lg.gather(1, top_i) - torch.from_numpy(a["long_top_val"][:n])).abs().max())
    assert err < 5e-4, err
    clear = torch.from_numpy(a["long_margin"][:n]) > 1e-3
    assert torch.equal(lg.argmax(dim=1)[clear], toks_all[:n][clear])
    sel = [(i, int(s)) for i, s in enumerate(ds) if int(s) < n]
    for i, s in sel:
        e = max(float((lg[s][torch.from_numpy(a["long_dense_top_idx"][i].astype(np.int64))] - torch.from_numpy(a["long_dense_top_val"][i])).abs().max()),
                float((lg[s][torch.from_numpy(a["long_dense_cols"].astype(np.int64))] - torch.from_numpy(a["long_dense_logits_cols"][i])).abs().max()))
        assert e < 5e-4, (s, e)
