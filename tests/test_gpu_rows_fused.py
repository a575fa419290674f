"""The rows-looped two-launch decoder layer for batches of 2 .. 8 rows (csrc/experimental/rows_fused.hpp, option rows_fused = 1; NOT the default: it is
bit-identical to batch-1 runs and needs 51 launches per step instead of 125, but the per-row work serialised inside each block makes it
1.4-1.9x slower than the matrix-core launch chain -- profiles/r03_rows_fused_*.txt) at the 350M shape: every block keeps its
weight rows in registers and loops over the batch rows, the in-launch all-gathers carry all rows at once, and the arithmetic per row
is the batch-1 chain's -- so the criterion is not a tolerance: row b of a batch must produce EXACTLY the tokens and the final logits
of its own batch-1 run (which tests/test_gpu_persist.py ties bit for bit to the five-launch chain).  BASELINE.json configs 3-5 in
small: meshanything.py:143-162 with a batch of rows."""
import os

import numpy as np
import pytest
import torch

from meshanything_amd.config import MAConfig, DTYPE_BF16
from conftest import load_weights_cached
from conftest import mouse_variants

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(golden_dir):
    from meshanything_amd.engine import Engine
    cfg = MAConfig.full(dtype=DTYPE_BF16, max_batch=8)
    e = Engine(cfg)
    if not e.get_option("experimental"):
        pytest.skip("the rows-looped launches are not part of the product build (build and run with MA_EXPERIMENTAL=1)")
    load_weights_cached(e, cfg, init="diverse")
    e.set_option("gemm_splitk", 0)                  # (the subject here is the decode form: keep the prefill's fc2 in ONE sum along K at every batch size -- at 8 .. ~38
                                                    #  samples it is otherwise four partial sums, csrc/gemm256.hpp GemmSplitK, and a batch's rows differ from their batch-1 runs in the last bits)
    e.set_option("rows_fused", 1)                   # opt-in path (measured slower than the matrix-core chain; kept for its bit-identity property)
    if e.get_option("rows_fused") != 1:
        pytest.skip("the rows-looped launches are not available on this device (need 256 CUs with 130 KB of LDS each)")
    x = mouse_variants(golden_dir, 8)
    _, e.prefix = e.encode(x.cuda())
    yield e
    e.close()


def _gen(eng, prefix, **kw):
    toks, lengths = eng.generate(prefix, **kw)
    lg = torch.stack([eng.read_logits(r).clone() for r in range(prefix.shape[0])])
    torch.cuda.synchronize()
    return toks.cpu(), list(lengths), lg.cpu()


@pytest.mark.parametrize("B", [8, 5, 4, 3, 2])
def test_batch_rows_are_bitwise_their_batch1_runs(eng, B):
    n = 96
    eng.set_option("rows_fused_min", 2)
    try:
        toks, lens, lg = _gen(eng, eng.prefix[:B], max_new_tokens=n, suppress_eos=True)
    finally:
        eng.set_option("rows_fused_min", 4)
    assert toks.shape == (B, n)
    for b in range(B):
        one, _, l1 = _gen(eng, eng.prefix[b:b + 1], max_new_tokens=n, suppress_eos=True)
        assert torch.equal(one[0], toks[b]), f"row {b} of a batch of {B}: tokens differ from the batch-1 run at step {int((one[0] != toks[b]).nonzero()[0])}"
        assert torch.equal(l1[0].view(torch.int32), lg[b].view(torch.int32)), f"row {b} of a batch of {B}: final logits differ by {float((l1[0] - lg[b]).abs().max()):.3e}"


def test_rows_fused_semantics_and_fallbacks(eng):
    """eos / pad bookkeeping, sampling with injected uniforms, determinism, graph == eager; the matrix-core path (rows_fused = 0) stays
    available and agrees except at near-ties."""
    B, n = 8, 160
    pre = eng.prefix
    a, la, _ = _gen(eng, pre, max_new_tokens=n, check_every=7)                      # natural eos
    for b in range(B):
        one, l1, _ = _gen(eng, pre[b:b + 1], max_new_tokens=n, check_every=7)
        m = int(l1[0])
        assert la[b] == m and torch.equal(a[b, :m], one[0, :m]) and (a[b, m:] == 2).all()
    again, _, _ = _gen(eng, pre, max_new_tokens=n, check_every=7)
    assert torch.equal(a, again)
    eng.set_option("use_graph", 0)
    try:
        eager, _, _ = _gen(eng, pre, max_new_tokens=64, suppress_eos=True)
    finally:
        eng.set_option("use_graph", 1)
    graph, _, _ = _gen(eng, pre, max_new_tokens=64, suppress_eos=True)
    assert torch.equal(eager, graph)
    u = torch.rand(B, 64, generator=torch.Generator().manual_seed(3))
    s, _, _ = _gen(eng, pre, sampling=True, uniforms=u, max_new_tokens=64, suppress_eos=True)
    for b in (0, 3, 7):
        one, _, _ = _gen(eng, pre[b:b + 1], sampling=True, uniforms=u[b:b + 1], max_new_tokens=64, suppress_eos=True)
        assert torch.equal(one[0], s[b])
    eng.set_option("rows_fused", 0)
    try:
        assert eng.get_option("rows_fused") == 0
        mfma, _, _ = _gen(eng, pre, max_new_tokens=64, suppress_eos=True)
    finally:
        eng.set_option("rows_fused", 1)
    same = int((mfma == graph).all(dim=1).sum())
    print(f"[rows fused] 8 rows x 64 tokens: {same}/8 rows token-identical to the matrix-core decode path (different summation order)")
    assert same >= 1                 # (a report: two summation orders fork at the first near-tie; every row is verified by the oracle elsewhere)


def test_rows_fused_step_timing_report(eng):
    """Report-only: the decode step of a batch (graph replay) on the rows-looped launches and on the matrix-core launch chain."""
    for B in (4, 8):
        eng.set_option("profile_batch", B)
        for L in (300, 3858, eng.cfg.max_seq - 120):
            row = {}
            for rf in (0, 1, 0, 1):
                eng.set_option("rows_fused", rf)
                eng.profile_decode(L, 2)
                p = eng.profile_decode(L, 12)
                row[rf] = (p["step_ms_graph"] * 1e3, sum(p["launches"].values()) // 12)
            eng.set_option("rows_fused", 1)
            print(f"[rows fused A/B] B {B} kv_len {L:5d}: matrix-core chain {row[0][0]:7.1f} us/step ({row[0][1]} launches) | rows-looped {row[1][0]:7.1f} us/step "
                  f"({row[1][1]} launches) | ratio {row[1][0] / row[0][0]:.3f} | {B / row[1][0] * 1e6:.0f} tok/s")
    eng.set_option("profile_batch", 1)
