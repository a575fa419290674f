"""The one-launch first half of a decoder layer for 8 rows (csrc/rows_attn.hpp: LayerNorm 2 + q/k/v + two-block attention + out_proj; VERDICT r4
item 2) against the three launches it replaces, at the 350M shape.  It runs the same arithmetic in the same order (the LayerNorm and q/k/v
arithmetic of gemm_dec_ln_kernel<.., 8>, the rounds and merges of attn_decode_final_kernel<8, true>, the K split of gemm_dec_kernel<1, 8>), so
the test demands BIT-IDENTICAL logits on every step, not a tolerance; the parity of those kernels against the reference's numbers
(test_gpu_reference_anchor.py, test_gpu_long_context.py at 8 rows) then carries over -- and those tests run the fused launch by default."""
import numpy as np
import pytest
import torch

from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F16
from conftest import load_weights_cached, mouse_variants

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["bf16", "fp16"])
def eng8(request, golden_dir):
    from meshanything_amd.engine import Engine
    cfg = MAConfig.full(dtype=DTYPE_BF16 if request.param == "bf16" else DTYPE_F16, max_batch=8)
    e = Engine(cfg)
    load_weights_cached(e, cfg, init="diverse")
    _, e.prefix = e.encode(mouse_variants(golden_dir, 8).cuda())
    e.policy = request.param
    return e


def _launches(eng, kv=600):
    eng.set_option("profile_batch", 8)
    p = eng.profile_decode(kv, 2)
    return sum(p["launches"].values()), p


def test_fused_first_half_is_bitwise_the_three_launches(eng8):
    if eng8.get_option("chain_resident") != 1:
        pytest.skip("the fused launches are not in use on this device")
    assert eng8.get_option("fuse_rows_attn") == 1
    n = 1200                                                 # cache to 1 456 positions: six rounds of 256, both register sets re-issued

    def run(fuse, **kw):
        eng8.set_option("fuse_rows_attn", fuse)
        try:
            return eng8.generate(eng8.prefix, max_new_tokens=n, suppress_eos=True, return_logits=True, **kw)
        finally:
            eng8.set_option("fuse_rows_attn", 1)
    t1, l1, g1 = run(1)
    t0, l0, g0 = run(0)
    assert t1.shape == (8, n) and len({tuple(r.tolist()) for r in t1.cpu()}) == 8 and len(set(t1[0].tolist())) > 64
    same = torch.equal(g0.view(torch.int32), g1.view(torch.int32))
    if not same:
        d = (g0 - g1).abs()
        step = int((d.amax(dim=(0, 2)) > 0).nonzero()[0])
        raise AssertionError(f"logits differ from step {step} on: max abs {float(d.max()):.3e}; rows differing at that step {(d[:, step].amax(dim=1) > 0).nonzero().flatten().tolist()}")
    assert torch.equal(t0, t1)
    again, _, g2 = run(1)
    assert torch.equal(again, t1) and torch.equal(g2.view(torch.int32), g1.view(torch.int32)), "the fused launch is not deterministic"
    del g0, g1, g2
    # sampling + teacher forcing go through the same launches
    u = torch.rand(8, 64, generator=torch.Generator().manual_seed(3))
    eng8.set_option("fuse_rows_attn", 0)
    s0, _ = eng8.generate(eng8.prefix, max_new_tokens=64, suppress_eos=True, sampling=True, uniforms=u)
    eng8.set_option("fuse_rows_attn", 1)
    s1, _ = eng8.generate(eng8.prefix, max_new_tokens=64, suppress_eos=True, sampling=True, uniforms=u)
    assert torch.equal(s0, s1)
    # eager launches == graph replay
    eng8.set_option("use_graph", 0)
    try:
        e1, _ = eng8.generate(eng8.prefix, max_new_tokens=96, suppress_eos=True)
    finally:
        eng8.set_option("use_graph", 1)
    assert torch.equal(e1, t1[:, :96])


def test_fused_first_half_launch_count_and_step_time(eng8):
    """76 launches per step at 8 rows instead of 124 (embedding, 24 x 3, the last LayerNorm, lm_head, pick); A/B of the step at mid cache."""
    if eng8.get_option("chain_resident") != 1:
        pytest.skip("the fused launches are not in use on this device")
    rows = []
    for kv in (600, 3858, 7300):
        eng8.set_option("fuse_rows_attn", 0)
        n0, p0 = _launches(eng8, kv)
        eng8.set_option("fuse_rows_attn", 1)
        n1, p1 = _launches(eng8, kv)
        rows.append((kv, n0, p0["step_ms_graph"], n1, p1["step_ms_graph"]))
        print(f"[8 rows, {eng8.policy}, kv {kv:5d}] five launches per layer: {n0 // 2} launches, {1e3 * p0['step_ms_graph']:7.1f} us/step | fused first half: {n1 // 2} launches, "
              f"{1e3 * p1['step_ms_graph']:7.1f} us/step | ratio {p1['step_ms_graph'] / p0['step_ms_graph']:.3f}")
    assert all(r[3] // 2 <= 76 for r in rows), rows
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
    side = torch.cuda.Stream()
    release = torch.zeros(1, dtype=torch.int32).pin_memory()
    torch.cuda.synchronize()
    eng8.occupy_cus(224, 2_000_000, stream=side, release=release)
    try:
        with pytest.warns(RuntimeWarning, match="fused decode launches timed out"):
            got, got_len = eng8.generate(eng8.prefix, max_new_tokens=n, suppress_eos=True)
    finally:
        release[0] = 1
    torch.cuda.synchronize()
    try:
        assert eng8.get_option("chain_fallbacks") == base + 1, "the starved grid was not noticed"
        assert eng8.get_option("chain_resident") == 0
        assert torch.equal(got.cpu(), want) and list(got_len) == list(want_len)
    finally:
        side.synchronize()
        eng8.set_option("chain_resident", 1)
    back, _ = eng8.generate(eng8.prefix, max_new_tokens=n, suppress_eos=True)
    assert back.shape == (8, n)
