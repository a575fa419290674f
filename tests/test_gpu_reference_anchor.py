"""Engine-independent anchors on MI355X: the HIP engine -- above all its bf16 policy, the one the benchmark runs,
whose own parity tests compare against an oracle that mirrors its rounding points -- against numbers produced by the REFERENCE's
code (tests/golden/make_golden.py, committed fixtures; nothing here reads /root/reference):

  * tiny shape: tokens of the reference's own `ShapeOPT` CausalLM wrapper under HuggingFace `GenerationMixin.generate`
    (meshanything.py:143-151) -- full length, truncated, and with rows reaching eos at different steps;
  * 350M shape, pc_examples/mouse.npy: per-step logits (ma_engine_read_logits) of a 65-token greedy decode against the reference's
    ShapeOPTDecoder.forward trace, within a stated bound, and the same token wherever the reference's top-1/top-2 margin exceeds
    twice that bound; the detokenizer's bins against NoiseResistantDecoder.forward's wherever ITS margin exceeds the bound.
Bounds (max abs logit error against the fp32 reference): fp32 policy 2e-3; bf16 policy 6e-2 for the decoder (24 layers of bf16 GEMV
inputs) and 2.5e-2 for the detokenizer logits (measured: 0.036 and 0.015, profiles/r03_reference_anchor.txt).

  * 350M shape, DIVERSE token streams (full_anchor_hf.npz; round 4): the fixture above walks a constant stream (token 2668 x 65: the default
    synthetic checkpoint has a fixed point), so every token-dependent path saw one input.  `test_350m_logits_along_the_reference_path`
    TEACHER-FORCES the engine (ma_sample_cfg.forced_tokens + logits_out) along two reference-produced paths of 257 tokens with
    >= 32 distinct ids -- a greedy one (weights init="diverse") and one drawn from the reference's own top-k/top-p distribution with
    stored uniforms (HF-style weights) -- and compares the logits of EVERY step (no early break at a fork), the argmax wherever the
    reference's margin is decisive, the sampler's draws, the encoder's activations and the detokenizer's bins on those weights."""
import os

import numpy as np
import pytest
import torch

from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F16, DTYPE_F32
from conftest import cached_state_dict, fused_generate, load_weights_cached, oracle_device

pytestmark = pytest.mark.gpu
POLICIES = {"fp32": DTYPE_F32, "bf16": DTYPE_BF16, "fp16": DTYPE_F16}


@pytest.mark.parametrize("policy", ["fp32", "bf16", "fp16"])
def test_tokens_of_the_reference_shapeopt_under_hf_generate(policy, golden_dir):
    from meshanything_amd.engine import Engine
    from oracle.meshanything_oracle import Oracle, verify_greedy_stream
    g = dict(np.load(os.path.join(golden_dir, "shapeopt_generate.npz")))
    cfg = MAConfig.tiny(dtype=POLICIES[policy], max_batch=4)
    sd = synthetic_state_dict(cfg)
    prefix = torch.from_numpy(g["gen_prefix"])
    tol = {"fp32": 2e-4, "bf16": 2e-2, "fp16": 3e-3}[policy]

    def check(sd_, want, **kw):
        eng = Engine(cfg)
        eng.load_weights(sd_.items())
        toks, lengths = eng.generate(prefix.cuda(), **kw)
        toks = toks.cpu()
        want = torch.from_numpy(want)
        if torch.equal(toks, want):
            return 0
        # a near-tie may resolve differently (summation order; bf16 rounding): then the engine's stream must still be a valid
        # greedy decode under the oracle -- which the CPU suite pins to these very fixtures -- and rows without a near-tie must match
        ora = Oracle(cfg, sd_, policy, device=oracle_device())
        differ = 0
        for b in range(prefix.shape[0]):
            n = int(lengths[b])
            v = verify_greedy_stream(ora, prefix[b:b + 1], toks[b, :n], tol)
            assert v["hard"] == [], (b, v)
            if v["ambiguous"] == 0:
                m = min(n, want.shape[1])
                assert torch.equal(toks[b, :m], want[b, :m]), f"row {b} differs from the reference's tokens without any near-tie"
            else:
                differ += 1
        return differ
    d = check(sd, g["gen_tokens"])
    d += check(sd, g["gen_tokens_max11"], max_new_tokens=11)
    tok = int(g["gen_eos_swap_token"][0])
    sd2 = dict(sd)
    w = sd["transformer.lm_head.weight"].copy()
    w[[1, tok]] = w[[tok, 1]]
    sd2["transformer.lm_head.weight"] = w
    d += check(sd2, g["gen_tokens_eos"], check_every=3)
    print(f"[{policy}] rows that left the reference's token stream at a near-tie: {d} of 12")
    assert d <= (0 if policy == "fp32" else 4)


@pytest.mark.parametrize("policy", ["bf16", "fp32"])
def test_350m_logits_and_bins_against_the_reference_modules(policy, golden_dir):
    from meshanything_amd.engine import Engine
    a = dict(np.load(os.path.join(golden_dir, "full_anchor.npz")))
    d = dict(np.load(os.path.join(golden_dir, "dataset.npz")))
    full = dict(np.load(os.path.join(golden_dir, "full.npz")))
    cfg = MAConfig.full(dtype=POLICIES[policy], max_batch=1)
    eng = Engine(cfg)
    load_weights_cached(eng, cfg)
    x = torch.from_numpy(d["mouse_norm"])[None].cuda()
    lat, prefix = eng.encode(x)                                   # the engine's own encoder: the whole chain is under test
    bound = {"fp32": 2e-3, "bf16": 6e-2}[policy]
    ref_tok = a["anchor_tokens"]
    top_i = torch.from_numpy(a["anchor_top_idx"]).long().cuda()
    top_v = torch.from_numpy(a["anchor_top_val"]).cuda()
    cols = torch.from_numpy(a["anchor_cols"]).long().cuda()
    lcols = torch.from_numpy(a["anchor_logits_cols"]).cuda()
    worst, compared = 0.0, 0
    for n in range(1, len(ref_tok) + 1):
        toks, _ = eng.generate(prefix, max_new_tokens=n, suppress_eos=True)
        toks = toks[0].cpu().numpy()
        j = n - 1                                                  # read_logits = the logits that chose token j
        if not np.array_equal(toks[:j], ref_tok[:j]):
            break                                                  # the context differs from the reference's: nothing to compare any more
        lg = eng.read_logits(0)
        err = max(float((lg[top_i[j]] - top_v[j]).abs().max()), float((lg[cols] - lcols[j]).abs().max()))
        worst = max(worst, err)
        compared += 1
        assert err <= bound, f"step {j}: logits differ from the reference's by {err:.4f} (bound {bound})"
        if a["anchor_margin"][j] > 2 * bound:
            assert toks[j] == ref_tok[j], f"step {j}: token {toks[j]} != reference {ref_tok[j]} at a reference margin of {a['anchor_margin'][j]:.4f}"
    print(f"[{policy}] 350M decoder vs the reference's ShapeOPTDecoder.forward: {compared} steps compared, max abs logit error {worst:.5f} "
          f"(bound {bound}); reference top-1/top-2 margin min {a['anchor_margin'].min():.4f}")
    assert compared == len(ref_tok), "the engine left the reference's greedy path although every margin exceeds twice the bound"
    # detokenizer: bins of the golden ids on the engine's own latents against the reference's bins and margins
    ids = torch.from_numpy(full["full_detok_ids"]).cuda()
    coords = eng.detokenize(ids, lat).cpu()
    valid = torch.from_numpy(a["anchor_detok_valid"])
    assert torch.equal(~torch.isnan(coords[0, :, 0, 0]), valid)
    bins = torch.round((coords[0].reshape(-1, 9) + 0.5) * cfg.discrete_num).long()           # undiscretize^-1 (meshanything.py:214-223)
    ref_bins = torch.from_numpy(a["anchor_detok_bins"]).long()
    margin = torch.from_numpy(a["anchor_detok_margin"])
    dbound = {"fp32": 2e-3, "bf16": 2.5e-2}[policy]
    diff = (bins != ref_bins) & valid[:, None]
    n_valid = int(valid.sum()) * 9
    print(f"[{policy}] 350M detokenizer vs the reference's NoiseResistantDecoder.forward: {int(diff.sum())} of {n_valid} bins differ; "
          f"largest reference margin among them {float(margin[diff].max()) if diff.any() else 0.0:.4f} (bound {2 * dbound})")
    assert not diff.any() or float(margin[diff].max()) <= 2 * dbound, "a bin differs where the reference's own top-1/top-2 margin is decisive"
    assert int(diff.sum()) <= 0.05 * n_valid
    eng.close()


# ---- diverse streams: teacher-forced along the reference's path ---------------------------------------------------------------------------
# bound = max abs logit error against the reference's fp32 numbers on ALL 257 steps of its path
# (measured on MI355X, profiles/r04_*: fp32 1e-5, bf16 0.021, fp16 see the report)
PATH_BOUND = {"fp32": 2e-3, "bf16": 8e-2, "fp16": 1e-2}


@pytest.mark.parametrize("tag,init", [("dva", "diverse"), ("hfa", "hf")])
@pytest.mark.parametrize("policy", ["bf16", "fp16", "fp32"])
def test_350m_logits_along_the_reference_path(policy, tag, init, golden_dir):
    from meshanything_amd.engine import Engine
    a = dict(np.load(os.path.join(golden_dir, "full_anchor_hf.npz")))
    d = dict(np.load(os.path.join(golden_dir, "dataset.npz")))
    cfg = MAConfig.full(dtype=POLICIES[policy], max_batch=1)
    eng = Engine(cfg)
    load_weights_cached(eng, cfg, init=init)
    x = torch.from_numpy(d["mouse_norm"])[None].cuda()
    lat, prefix = eng.encode(x)                                   # the engine's own encoder: the whole chain is under test
    rows = a[f"{tag}_rows"]
    e_lat = max(float(np.abs(lat[0, rows].cpu().numpy() - a[f"{tag}_latents_rows"]).max()), float(np.abs(lat[0, :, :8].cpu().numpy() - a[f"{tag}_latents_cols8"]).max()))
    e_pre = max(float(np.abs(prefix[0, rows].cpu().numpy() - a[f"{tag}_prefix_rows"]).max()), float(np.abs(prefix[0, :, :8].cpu().numpy() - a[f"{tag}_prefix_cols8"]).max()))
    ref_tok = a[f"{tag}_tokens"]
    n = len(ref_tok)
    assert len(set(ref_tok.tolist())) >= 32, "the anchor stream is not diverse"
    sampled = int(a[f"{tag}_mode"][0]) == 1
    forced = torch.from_numpy(ref_tok)[None]
    u = torch.from_numpy(a[f"{tag}_uniforms"])[None]
    # (16-bit policies: the two fused launches per layer of the batch-1 chain)
    toks, lengths, logits = fused_generate(eng, f"{policy} reference anchor, batch 1", prefix, max_new_tokens=n, suppress_eos=True, forced_tokens=forced, return_logits=True,
                                             sampling=sampled, uniforms=u if sampled else None)
    assert toks.shape == (1, n) and int(lengths[0]) == n
    lg = logits[0]
    top_i = torch.from_numpy(a[f"{tag}_top_idx"]).long().cuda()
    top_v = torch.from_numpy(a[f"{tag}_top_val"]).cuda()
    cols = torch.from_numpy(a[f"{tag}_cols"]).long().cuda()
    lcols = torch.from_numpy(a[f"{tag}_logits_cols"]).cuda()
    margin = torch.from_numpy(a[f"{tag}_margin"]).cuda()
    err = torch.maximum((lg.gather(1, top_i) - top_v).abs().max(dim=1).values, (lg[:, cols] - lcols).abs().max(dim=1).values)
    bound = PATH_BOUND[policy]
    lg2 = lg.clone()
    lg2[:, 1] = float("-inf")
    eng_arg = lg2.argmax(dim=1)
    ref_arg = top_i[:, 0]
    agree = float((eng_arg == ref_arg).float().mean())
    decisive = margin > 2 * bound
    picks = toks[0]
    draw_equal = float((picks.cpu() == torch.from_numpy(ref_tok)).float().mean())
    print(f"[{policy}/{tag}] {n} steps along the reference's {'sampled' if sampled else 'greedy'} path ({len(set(ref_tok.tolist()))} distinct ids): max abs logit error "
          f"{float(err.max()):.5f} (bound {bound}), median {float(err.median()):.5f}; argmax agreement {agree * 100:.2f} % ({int(decisive.sum())} decisive steps, reference "
          f"margin median {float(margin.median()):.4f}); engine's own {'draws' if sampled else 'picks'} equal the reference's tokens at {draw_equal * 100:.2f} % of the steps; "
          f"encoder latents err {e_lat:.3e}, prefix err {e_pre:.3e}")
    assert float(err.max()) <= bound, f"step {int(err.argmax())}: logits differ from the reference's by {float(err.max()):.4f}"
    if policy == "fp16":
        # the fixture's second set of numbers: the reference's OWN modules under torch.autocast(float16) along the same tokens -- the
        # reference's real precision class (make_golden.py: golden_anchor_diverse).  Two fp16 implementations differ from each other by
        # about what each differs from fp32.
        err16 = torch.maximum((lg.gather(1, top_i) - torch.from_numpy(a[f"{tag}_f16_top_val"]).cuda()).abs().max(dim=1).values,
                              (lg[:, cols] - torch.from_numpy(a[f"{tag}_f16_logits_cols"]).cuda()).abs().max(dim=1).values)
        ref16 = torch.from_numpy(a[f"{tag}_f16_argmax"]).cuda()
        print(f"[fp16/{tag}] against the reference under fp16 autocast: max abs logit error {float(err16.max()):.5f}, median {float(err16.median()):.5f}; "
              f"argmax agreement {float((eng_arg == ref16).float().mean()) * 100:.2f} %")
        assert float(err16.max()) <= 1.5e-2
    assert bool((eng_arg[decisive] == ref_arg[decisive]).all()), "argmax differs from the reference's at a decisive margin"
    if not sampled:
        assert torch.equal(picks, eng_arg), "the pick kernel's token is not the argmax of the logits it returned"
        assert agree >= {"fp32": 0.999, "fp16": 0.99, "bf16": 0.80}[policy]
    else:
        # the sampler on the reference's own context, with the reference's uniforms, against transformers' warpers: fp32 may differ where
        # a CDF edge or the top-p cut sits within rounding of the uniform; bf16 logits move the edges by a few 1e-2
        assert draw_equal >= {"fp32": 0.97, "fp16": 0.85, "bf16": 0.40}[policy], draw_equal
        assert int(picks.min()) >= 0 and not bool((picks == 1).any())
    assert e_lat < 1e-5 and e_pre < 1e-4                            # the encoder is exact under the bf16 policy too (cfg.enc_exact)
    # detokenizer on these weights: bins against the reference's wherever ITS margin is decisive
    full = dict(np.load(os.path.join(golden_dir, "full.npz")))
    ids = torch.from_numpy(full["full_detok_ids"]).cuda()
    coords = eng.detokenize(ids, lat).cpu()
    valid = torch.from_numpy(a[f"{tag}_detok_valid"])
    assert torch.equal(~torch.isnan(coords[0, :, 0, 0]), valid)
    bins = torch.round((coords[0].reshape(-1, 9) + 0.5) * cfg.discrete_num).long()
    ref_bins = torch.from_numpy(a[f"{tag}_detok_bins"]).long()
    dmargin = torch.from_numpy(a[f"{tag}_detok_margin"])
    dbound = {"fp32": 2e-3, "bf16": 2.5e-2, "fp16": 5e-3}[policy]
    diff = (bins != ref_bins) & valid[:, None]
    print(f"[{policy}/{tag}] detokenizer: {int(diff.sum())} of {int(valid.sum()) * 9} bins differ; largest reference margin among them "
          f"{float(dmargin[diff].max()) if diff.any() else 0.0:.4f} (bound {2 * dbound})")
    assert not diff.any() or float(dmargin[diff].max()) <= 2 * dbound
    eng.close()


@pytest.mark.parametrize("B", [8, 16, 40, 64])
@pytest.mark.parametrize("policy", ["bf16", "fp16"])
def test_350m_batched_decode_along_the_reference_path(policy, B, golden_dir):
    """The batched decode path (skinny matrix-core GEMMs with / without the LayerNorm folded in, final-form attention in its two-block and
    one-block forms, launched row prologues, one to four 16-row batch tiles per block: 8, 16, 40 and 64 rows) against the REFERENCE's own logits: rows 0 .. B - 1 carry the anchor's cloud and
    are teacher-forced along the reference's greedy path (`dva`, 257 steps, 75 distinct ids); every row's logits stay within the policy's
    bound of the reference's on every step, and argmax equals the reference's wherever its margin is decisive.  (The batch-1 tests above
    cover the fused launches; this one holds the kernels that a `--gpus N` run executes to the same numbers.)"""
    from meshanything_amd.engine import Engine
    a = dict(np.load(os.path.join(golden_dir, "full_anchor_hf.npz")))
    d = dict(np.load(os.path.join(golden_dir, "dataset.npz")))
    tag = "dva"
    cfg = MAConfig.full(dtype=POLICIES[policy], max_batch=B)
    eng = Engine(cfg)
    load_weights_cached(eng, cfg, init="diverse")
    x = torch.from_numpy(d["mouse_norm"])[None].expand(B, -1, -1).contiguous().cuda()
    _, prefix = eng.encode(x)
    ref_tok = a[f"{tag}_tokens"]
    n = len(ref_tok)
    forced = torch.from_numpy(ref_tok)[None].expand(B, -1).contiguous()
    toks, lengths, logits = fused_generate(eng, f"{policy} reference anchor, batch {B}", prefix, max_new_tokens=n, suppress_eos=True, forced_tokens=forced, return_logits=True)
    assert toks.shape == (B, n) and all(int(l) == n for l in lengths)
    top_i = torch.from_numpy(a[f"{tag}_top_idx"]).long().cuda()
    top_v = torch.from_numpy(a[f"{tag}_top_val"]).cuda()
    cols = torch.from_numpy(a[f"{tag}_cols"]).long().cuda()
    lcols = torch.from_numpy(a[f"{tag}_logits_cols"]).cuda()
    margin = torch.from_numpy(a[f"{tag}_margin"]).cuda()
    bound = PATH_BOUND[policy]
    decisive = margin > 2 * bound
    worst, agree = 0.0, 1.0
    for b in range(B):
        lg = logits[b]
        err = torch.maximum((lg.gather(1, top_i) - top_v).abs().max(dim=1).values, (lg[:, cols] - lcols).abs().max(dim=1).values)
        worst = max(worst, float(err.max()))
        lg2 = lg.clone()
        lg2[:, 1] = float("-inf")
        arg = lg2.argmax(dim=1)
        agree = min(agree, float((arg == top_i[:, 0]).float().mean()))
        assert float(err.max()) <= bound, f"row {b}, step {int(err.argmax())}: logits differ from the reference's by {float(err.max()):.4f}"
        assert bool((arg[decisive] == top_i[decisive, 0]).all()), f"row {b}: argmax differs from the reference's at a decisive margin"
        assert torch.equal(toks[b], arg), "the pick kernel's token is not the argmax of the logits it returned"
    print(f"[{policy}/batch {B}] matrix-core decode path, {B} rows x {n} steps along the reference's greedy path: max abs logit error {worst:.5f} "
          f"(bound {bound}); lowest argmax agreement of a row {agree * 100:.2f} %")
    eng.close()


def test_forced_tokens_walk_the_given_stream_tiny():
    """ma_sample_cfg.forced_tokens / logits_out on the tiny shape, every decode path (batch 1 chain, rows in the grid, matrix-core batch):
    forcing the engine's OWN greedy stream reproduces it and its logits; forcing another stream reports, at every step, the argmax of
    the oracle's teacher-forced distribution for that stream."""
    from meshanything_amd.engine import Engine
    from oracle.meshanything_oracle import Oracle
    for policy, B in (("fp32", 1), ("fp32", 3), ("bf16", 1), ("bf16", 4), ("fp16", 2), ("fp16", 4)):
        cfg = MAConfig.tiny(dtype=POLICIES[policy], max_batch=4)
        sd = synthetic_state_dict(cfg)
        eng = Engine(cfg)
        eng.load_weights(sd.items())
        ora = Oracle(cfg, sd, policy, device=oracle_device())
        g = torch.Generator().manual_seed(17 + B)
        prefix = torch.randn(B, cfg.cond_length, cfg.hidden, generator=g) * 0.7
        n = cfg.max_new_tokens
        free, _, lg_free = eng.generate(prefix.cuda(), suppress_eos=True, return_logits=True)
        again, _, lg_again = eng.generate(prefix.cuda(), suppress_eos=True, forced_tokens=free, return_logits=True)
        assert torch.equal(free, again) and torch.equal(lg_free, lg_again), "forcing the engine's own stream changed it"
        other = torch.randint(3, cfg.vocab, (B, n), generator=g)
        other[:, 0] = 0
        picks, lengths, lg = eng.generate(prefix.cuda(), suppress_eos=True, forced_tokens=other, return_logits=True)
        assert picks.shape == (B, n) and (lengths == n).all()
        tol = {"fp32": 2e-4, "bf16": 3e-2, "fp16": 4e-3}[policy]
        for b in range(B):
            ref = ora.teacher_forced_logits(prefix[b:b + 1], other[b])[:n]
            e = float((lg[b].cpu() - ref).abs().max())
            assert e < tol, (policy, B, b, e)
            ref2 = ref.clone(); ref2[:, 1] = float("-inf")
            top2 = torch.topk(ref2, 2, dim=-1)
            clear = (top2.values[:, 0] - top2.values[:, 1]) > 2 * tol
            assert torch.equal(picks[b].cpu()[clear], top2.indices[:, 0][clear]), (policy, B, b)
        eng.close()
