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
