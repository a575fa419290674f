"""The two-launch decoder layer for 8 rows (csrc/rows_attn.hpp: [LayerNorm 2 +] q/k/v + two-block attention + out_proj; csrc/rows_mlp.hpp:
LayerNorm 1 + fc1 + fc2 [+ LayerNorm 2]) against the five launches it replaces, at the 350M shape.  Both run the same
arithmetic in the same order (the LayerNorm and q/k/v arithmetic of gemm_dec_ln_kernel<.., 8>, the rounds and merges of
attn_decode_final_kernel<8, true>, the K splits of gemm_dec_kernel<1, 8>), so the tests demand BIT-IDENTICAL logits on every step, not a
tolerance; the parity of those kernels against the reference's numbers (test_gpu_reference_anchor.py, test_gpu_long_context.py at 8 rows)
then carries over -- and those tests run the fused launches by default."""
import numpy as np
import pytest
import torch

from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F16
from conftest import generate_on_a_starved_device, load_weights_cached, mouse_variants

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["bf16", "fp16"])
def eng8(request, golden_dir):
    from meshanything_amd.engine import Engine
    cfg = MAConfig.full(dtype=DTYPE_BF16 if request.param == "bf16" else DTYPE_F16, max_batch=8)
    e = Engine(cfg)
    load_weights_cached(e, cfg, init="diverse")
    _, e.prefix = e.encode(mouse_variants(golden_dir, 8).cuda())
    e.policy = request.param
    yield e
    e.close()


def _launches(eng, kv=600):
    eng.set_option("profile_batch", 8)
    p = eng.profile_decode(kv, 2)
    return sum(p["launches"].values()), p


FORMS = {"five launches per layer": dict(fuse_rows_attn=0, fuse_rows_mlp=0, rows_mlp_ln2=1),
         "fused first half": dict(fuse_rows_attn=1, fuse_rows_mlp=0, rows_mlp_ln2=1),
         "fused second half (LayerNorm 2 left to the next launch)": dict(fuse_rows_attn=0, fuse_rows_mlp=1, rows_mlp_ln2=0),
         "both halves, LayerNorm 2 in the next launch": dict(fuse_rows_attn=1, fuse_rows_mlp=1, rows_mlp_ln2=0),
         "two launches per layer (default)": dict(fuse_rows_attn=1, fuse_rows_mlp=1, rows_mlp_ln2=1)}
DEFAULT = FORMS["two launches per layer (default)"]


def _set(eng, form):
    for k, v in form.items():
        eng.set_option(k, v)


def test_fused_halves_are_bitwise_the_five_launches(eng8):
    """Every combination of the two fused launches (and of where LayerNorm 2 runs) against the five-launch layer: bit-identical logits on all
    1 200 steps (cache to 1 456 positions: six rounds of 256 per (row, head), both register sets re-issued), for 8 distinct clouds."""
    if eng8.get_option("chain_resident") != 1:
        pytest.skip("the fused launches are not in use on this device")
    assert all(eng8.get_option(k) == v for k, v in DEFAULT.items())
    n = 1200

    def run(form, **kw):
        _set(eng8, form)
        try:
            return eng8.generate(eng8.prefix, max_new_tokens=n, suppress_eos=True, return_logits=True, **kw)
        finally:
            _set(eng8, DEFAULT)
    t0, l0, g0 = run(FORMS["five launches per layer"])
    assert t0.shape == (8, n) and len({tuple(r.tolist()) for r in t0.cpu()}) == 8 and len(set(t0[0].tolist())) > 64
    for name, form in FORMS.items():
        if name == "five launches per layer":
            continue
        t1, l1, g1 = run(form)
        if not torch.equal(g0.view(torch.int32), g1.view(torch.int32)):
            d = (g0 != g1)
            step = int(d.any(dim=2).any(dim=0).nonzero()[0])
            raise AssertionError(f"{name}: logits differ from step {step} on: max abs at that step {float((g0[:, step] - g1[:, step]).abs().max()):.3e}, "
                                 f"rows differing there {d[:, step].any(dim=1).nonzero().flatten().tolist()}")
        assert torch.equal(t0, t1), name
        del g1
    again, _, g2 = run(DEFAULT)
    assert torch.equal(again, t0) and torch.equal(g2.view(torch.int32), g0.view(torch.int32)), "the fused launches are not deterministic"
    del g0, g2
    # sampling goes through the same launches
    u = torch.rand(8, 64, generator=torch.Generator().manual_seed(3))
    _set(eng8, FORMS["five launches per layer"])
    s0, _ = eng8.generate(eng8.prefix, max_new_tokens=64, suppress_eos=True, sampling=True, uniforms=u)
    _set(eng8, DEFAULT)
    s1, _ = eng8.generate(eng8.prefix, max_new_tokens=64, suppress_eos=True, sampling=True, uniforms=u)
    assert torch.equal(s0, s1)
    # eager launches == graph replay
    eng8.set_option("use_graph", 0)
    try:
        e1, _ = eng8.generate(eng8.prefix, max_new_tokens=96, suppress_eos=True)
    finally:
        eng8.set_option("use_graph", 1)
    assert torch.equal(e1, t0[:, :96])


def test_request_placements_and_the_scalar_sweep_are_bitwise_the_same(eng8):
    """Option rows_attn_early: where the first cache rounds are requested inside the first fused launch (0 .. 4: sweep of the q/k/v granules by
    vector loads of wave 0; 5: by scalar loads -- s_load_dwordx16 glc -- of the waves 0 .. 3 while the waves 4 .. 7 already stream the cache; 6, the
    default: 5 + the rounds that lie wholly below the newest position reduced without masks and override).  A placement decides when bytes move, never which: every form gives the same logits on every step, to a cache of 857 positions
    (four rounds per (row, head): both register sets re-issued in both blocks of a pair)."""
    if eng8.get_option("chain_resident") != 1:
        pytest.skip("the fused launches are not in use on this device")
    assert eng8.get_option("rows_attn_early") == 6
    n = 600
    try:
        ref = None
        # (0, 1, 2, 4 -- measured, not kept -- exist in MA_EXPERIMENTAL=1 builds only)
        for early in ((6, 0, 1, 2, 3, 4, 5) if eng8.get_option("experimental") else (6, 3, 5)):
            eng8.set_option("rows_attn_early", early)
            t, _, g = eng8.generate(eng8.prefix, max_new_tokens=n, suppress_eos=True, return_logits=True)
            if ref is None:
                ref = (t, g)
                continue
            assert torch.equal(ref[1].view(torch.int32), g.view(torch.int32)) and torch.equal(ref[0], t), f"rows_attn_early={early} differs from the default"
            del g
    finally:
        eng8.set_option("rows_attn_early", 6)
    assert eng8.get_option("xchg_timeouts") == 0


def test_fused_halves_launch_count_and_step_time(eng8):
    """51 launches per step at 8 rows instead of 124 (embedding, 24 x 2, lm_head, pick); A/B of the step at three cache depths."""
    if eng8.get_option("chain_resident") != 1:
        pytest.skip("the fused launches are not in use on this device")
    counts = {}
    for kv in (600, 3858, 7300):
        row = []
        for name, form in FORMS.items():
            _set(eng8, form)
            n, p = _launches(eng8, kv)
            counts[name] = n // 2
            row.append(f"{name}: {n // 2} launches, {1e3 * p['step_ms_graph']:.1f} us")
        _set(eng8, DEFAULT)
        print(f"[8 rows, {eng8.policy}, kv {kv:5d}] " + " | ".join(row))
    assert counts["two launches per layer (default)"] <= 52 and counts["fused first half"] <= 76, counts
    assert eng8.get_option("xchg_timeouts") == 0


def test_fused_first_half_falls_back_when_the_device_is_shared(eng8):
    """Its 256 blocks of 8 waves need every CU.  With 224 CUs held by another stream the first fused launch cannot become resident: one bounded
    sweep raises the error word, the engine re-runs the generation without the fused launches (`chain_resident` off: one attention block per
    (row, head), five launches per layer) and returns what that form returns -- no MA_ERR_HIP, no hang."""
    if eng8.get_option("chain_resident") != 1:
        pytest.skip("the fused launches are not in use on this device")
    n = 24
    eng8.set_option("chain_resident", 0)
    want, want_len = eng8.generate(eng8.prefix, max_new_tokens=n, suppress_eos=True)
    eng8.set_option("chain_resident", 1)
    want = want.cpu()
    base = eng8.get_option("chain_fallbacks")
    (got, got_len), _, _ = generate_on_a_starved_device(eng8, lambda: eng8.generate(eng8.prefix, max_new_tokens=n, suppress_eos=True))
    try:
        assert eng8.get_option("chain_fallbacks") == base + 1, "the starved grid was not noticed"
        assert eng8.get_option("chain_resident") == 0
        assert torch.equal(got.cpu(), want) and list(got_len) == list(want_len)
    finally:
        eng8.set_option("chain_resident", 1)
    back, _ = eng8.generate(eng8.prefix, max_new_tokens=n, suppress_eos=True)
    assert back.shape == (8, n)
