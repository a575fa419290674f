"""The persistent decode step (csrc/experimental/persist.hpp: one resident launch per step) against the launch chain it replaces, at the
BASELINE.json configs[1] shape (350M, bf16, batch 1, greedy): the two implementations run the same arithmetic in the same
order, so the test demands bit-identical logits and token-identical streams, not a tolerance."""
import os

import numpy as np
import pytest
import torch

from meshanything_amd.config import MAConfig, DTYPE_BF16
from conftest import cached_state_dict, generate_on_a_starved_device, load_weights_cached

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["bf16", "fp16", "fp32"])       # fp32: the "exact" policy runs the same two fused launches since round 6
def eng(request, golden_dir):
    from meshanything_amd.engine import Engine
    from meshanything_amd.config import DTYPE_F16, DTYPE_F32
    cfg = MAConfig.full(dtype={"bf16": DTYPE_BF16, "fp16": DTYPE_F16, "fp32": DTYPE_F32}[request.param], max_batch=2)
    e = Engine(cfg)
    load_weights_cached(e, cfg, init="diverse")           # a greedy stream that depends on its own tokens (checkpoint.py)
    d = dict(np.load(os.path.join(golden_dir, "dataset.npz")))
    x = torch.from_numpy(d["mouse_norm"])[None]
    _, prefix = e.encode(x.cuda())
    e.prefix = prefix
    yield e
    e.close()                                             # deterministic: a collector-timed ma_engine_destroy (hipFree = device-wide sync) in the middle of a later test stalls it


def _need_persist(eng):
    if not eng.get_option("experimental"):
        pytest.skip("the persistent decode step is not part of the product build (build and run with MA_EXPERIMENTAL=1)")
    if not eng.persist_available():
        pytest.skip("persistent decode step not available on this device (needs 256 CUs)")


def _gen(eng, impl, **kw):
    eng.set_option("decode_impl", impl)
    try:
        toks, lengths = eng.generate(eng.prefix, **kw)
        logits = eng.read_logits(0).clone()
    finally:
        eng.set_option("decode_impl", 0)
    torch.cuda.synchronize()
    return toks.cpu(), lengths, logits.cpu()


def test_persistent_step_is_bitwise_the_launch_chain(eng):
    _need_persist(eng)
    for n in (2, 3, 17, 64):                        # the last step's logits after n - 1 decode steps
        t0, l0, g0 = _gen(eng, 0, max_new_tokens=n, suppress_eos=True)
        t1, l1, g1 = _gen(eng, 1, max_new_tokens=n, suppress_eos=True)
        assert torch.equal(t0, t1), f"tokens differ within {n} steps: {t0.tolist()} vs {t1.tolist()}"
        same = torch.equal(g0.view(torch.int32), g1.view(torch.int32))
        assert same, f"logits of step {n - 1} differ: max abs {float((g0 - g1).abs().max()):.3e} at {int((g0 - g1).abs().argmax())}"


def test_persistent_generate_400_tokens_token_identical(eng):
    _need_persist(eng)
    n = int(os.environ.get("MA_TEST_GEN_TOKENS", "400"))
    t0, l0, _ = _gen(eng, 0, max_new_tokens=n, suppress_eos=True)
    t1, l1, _ = _gen(eng, 1, max_new_tokens=n, suppress_eos=True)
    assert t0.shape == (1, n) and torch.equal(t0, t1)
    again, _, _ = _gen(eng, 1, max_new_tokens=n, suppress_eos=True)
    assert torch.equal(t1, again), "the persistent step is not deterministic"
    # eos allowed: same stopping point, same eos / pad tail
    t0, l0, _ = _gen(eng, 0, max_new_tokens=200, check_every=7)
    t1, l1, _ = _gen(eng, 1, max_new_tokens=200, check_every=7)
    assert torch.equal(t0, t1) and list(l0) == list(l1)
    # graph replay == eager launches
    eng.set_option("use_graph", 0)
    try:
        e1, _, _ = _gen(eng, 1, max_new_tokens=64, suppress_eos=True)
    finally:
        eng.set_option("use_graph", 1)
    g1, _, _ = _gen(eng, 1, max_new_tokens=64, suppress_eos=True)
    assert torch.equal(e1, g1)


def test_persistent_falls_back_outside_its_envelope(eng):
    """Sampling and batches of more than one row keep using the launch chain (same results as with decode_impl 0)."""
    _need_persist(eng)
    u = torch.rand(1, 48, generator=torch.Generator().manual_seed(5))
    eng.set_option("decode_impl", 1)
    try:
        a, _ = eng.generate(eng.prefix, sampling=True, uniforms=u, max_new_tokens=48, suppress_eos=True)
        two = torch.cat([eng.prefix, eng.prefix])
        b, _ = eng.generate(two, max_new_tokens=48, suppress_eos=True)
    finally:
        eng.set_option("decode_impl", 0)
    a0, _ = eng.generate(eng.prefix, sampling=True, uniforms=u, max_new_tokens=48, suppress_eos=True)
    b0, _ = eng.generate(torch.cat([eng.prefix, eng.prefix]), max_new_tokens=48, suppress_eos=True)
    assert torch.equal(a, a0) and torch.equal(b, b0)


def test_persistent_step_timing_report(eng):
    """Report-only: decode step (graph replay) of both implementations at three cache lengths + the edge timeline."""
    _need_persist(eng)
    cfg = eng.cfg
    for L in (300, 3858, cfg.max_seq - 80):
        row = {}
        for impl in (0, 1):
            eng.set_option("decode_impl", impl)
            eng.profile_decode(L, 2)
            p = eng.profile_decode(L, 16)
            row[impl] = p["step_ms_graph"] * 1e3
        eng.set_option("decode_impl", 0)
        print(f"[persist A/B] kv_len {L:5d}: launch chain {row[0]:7.1f} us/step | persistent {row[1]:7.1f} us/step | ratio {row[1] / row[0]:.3f}")
    tr = eng.persist_trace(3858).astype(np.int64)
    t = (tr - tr[:, :1]) / 100.0                       # us since each workgroup's start
    ev = t[:, 1:-1].reshape(256, -1, 2)                # per edge: (sweep start, gather done)
    wait = ev[:, :, 1] - ev[:, :, 0]
    seq = ["qkv", "part", "a", "y1", "ffn"] + ["y2", "qkv", "part", "a", "y1", "ffn"] * (cfg.layers - 1) + ["y2", "arg"]
    assert len(seq) == wait.shape[1]
    agg = {}
    for i, k in enumerate(seq):
        agg.setdefault(k, []).append(np.median(wait[:, i]))
    print("[persist trace] median sweep time per edge kind (us, median over workgroups then mean over layers): " +
          ", ".join(f"{k} {np.mean(v):.2f}" for k, v in agg.items()))
    # compute wave 0 of every workgroup: 19 stamps per layer
    ct = eng.last_compute_trace.astype(np.int64)
    nl = cfg.layers
    ct = ct[:, :19 * nl].reshape(256, nl, 19) / 100.0
    labels = ["qkv:in+LN", "qkv:weights", "qkv:dots", "qkv:publish", "attn:partial out", "part gathered", "merged out",
              "oproj:in", "oproj:weights", "oproj:dots", "oproj:publish", "fc1:in+LN", "fc1:weights", "fc1:dots", "fc1:publish",
              "fc2:in", "fc2:weights", "fc2:dots", "fc2:publish"]
    d = np.diff(ct, axis=2)                              # (256, nl, 18) intervals inside a layer
    first = ct[:, 1:, 0] - ct[:, :-1, 18]                # y2 edge + LN of the next layer
    med = np.median(d[:, 1:, :], axis=(0, 1))
    print("[persist compute-wave timeline] median interval (us) ending at each point, layers 1..: y2 edge+LN %.2f | " % np.median(first) +
          " | ".join(f"{labels[i + 1]} {med[i]:.2f}" for i in range(18)))
    print(f"[persist compute-wave timeline] layer period {np.median(ct[:, 2:, 0] - ct[:, 1:-1, 0]):.2f} us")
    print(f"[persist trace] step span: {np.median(t[:, -1]):.1f} us (median workgroup), in sweeps {np.median(wait.sum(axis=1)):.1f} us")


def test_fused_qkv_attention_launch_is_bitwise_the_two_launches(eng):
    """The launch chain's fused q/k/v + attention launch (csrc/qkv_attn.hpp, default) against the two launches it replaces
    (fuse_qkv_attn = 0): same arithmetic, so bit-identical logits and identical tokens -- batch 1 and a batch of 2 rows."""
    # on a whole MI355X the fused launches are what runs (their 256 blocks are resident together with margin)
    assert eng.get_option("chain_resident") == 1 and eng.get_option("fuse_qkv_attn") == 1 and eng.get_option("fuse_oproj_fc1") == 1 and eng.get_option("fuse_fc2") == 1
    def run(fuse, prefix, n, opt="fuse_qkv_attn"):
        eng.set_option(opt, fuse)
        try:
            toks, _ = eng.generate(prefix, max_new_tokens=n, suppress_eos=True)
            lg = [eng.read_logits(r).clone() for r in range(prefix.shape[0])]
        finally:
            eng.set_option(opt, 1)
        torch.cuda.synchronize()
        return toks.cpu(), [x.cpu() for x in lg]
    # the second fused launch of the chain (csrc/oproj_fc1.hpp): out_proj + LayerNorm + fc1, same criterion
    for n in (2, 9, 130):
        t0, g0 = run(0, eng.prefix, n, "fuse_oproj_fc1")
        t1, g1 = run(1, eng.prefix, n, "fuse_oproj_fc1")
        assert torch.equal(t0, t1), (n, t0.tolist(), t1.tolist())
        assert torch.equal(g0[0].view(torch.int32), g1[0].view(torch.int32)), f"oproj+fc1: logits differ after {n} tokens: {float((g0[0] - g1[0]).abs().max()):.3e}"
    two_ = torch.cat([eng.prefix, eng.prefix.flip(1)])
    t0, g0 = run(0, two_, 40, "fuse_oproj_fc1")
    t1, g1 = run(1, two_, 40, "fuse_oproj_fc1")
    assert torch.equal(t0, t1) and all(torch.equal(g0[r].view(torch.int32), g1[r].view(torch.int32)) for r in range(2))
    for L in (300, 3858, eng.cfg.max_seq - 80):
        row = {}
        for fuse in (0, 1):
            eng.set_option("fuse_oproj_fc1", fuse)
            eng.profile_decode(L, 2)
            row[fuse] = eng.profile_decode(L, 16)["step_ms_graph"] * 1e3
        eng.set_option("fuse_oproj_fc1", 1)
        print(f"[fused oproj+fc1 A/B] kv_len {L:5d}: two launches {row[0]:7.1f} us/step | fused {row[1]:7.1f} us/step | ratio {row[1] / row[0]:.3f}")
    for n in (2, 9, 130):
        t0, g0 = run(0, eng.prefix, n)
        t1, g1 = run(1, eng.prefix, n)
        assert torch.equal(t0, t1), (n, t0.tolist(), t1.tolist())
        assert torch.equal(g0[0].view(torch.int32), g1[0].view(torch.int32)), f"logits differ after {n} tokens: {float((g0[0] - g1[0]).abs().max()):.3e}"
    two = torch.cat([eng.prefix, eng.prefix.flip(1)])
    t0, g0 = run(0, two, 40)
    t1, g1 = run(1, two, 40)
    assert torch.equal(t0, t1)
    for r in range(2):
        assert torch.equal(g0[r].view(torch.int32), g1[r].view(torch.int32))
    # timing, report only
    for L in (300, 3858, eng.cfg.max_seq - 80):
        row = {}
        for fuse in (0, 1):
            eng.set_option("fuse_qkv_attn", fuse)
            eng.profile_decode(L, 2)
            row[fuse] = eng.profile_decode(L, 16)["step_ms_graph"] * 1e3
        eng.set_option("fuse_qkv_attn", 1)
        print(f"[fused qkv+attn A/B] kv_len {L:5d}: two launches {row[0]:7.1f} us/step | fused {row[1]:7.1f} us/step | ratio {row[1] / row[0]:.3f}")


def test_fused_launches_fall_back_when_the_device_is_shared(eng):
    """The fused launches spin on granules written by other blocks of their grid: they need all 256 blocks resident.  Another
    stream that holds most CUs (here: 224 workgroups with 160 KiB of LDS each = 224 whole CUs, until the test lets go; at most 2 s)
    breaks that: the 32 CUs that are left cannot hold 256 blocks of 256 threads at these kernels' register budgets.  The engine must notice (one
    bounded 20 ms sweep, then every other sweep gives up at once), switch to the five-launch chain -- which needs no co-residency
    and produces the same bits -- run the generation again and return the SAME tokens, not MA_ERR_HIP."""
    if eng.get_option("chain_resident") != 1:
        pytest.skip("the fused launches are not in use on this device")
    n = 24
    want, want_len = eng.generate(eng.prefix, max_new_tokens=n, suppress_eos=True)
    want = want.cpu()
    base = eng.get_option("chain_fallbacks")
    # 224 of the 256 CUs are gone until the helper lets go: the prefill runs on the 32 that are left (1/8 of the chip), the first fused decode
    # launch then cannot get its 256 blocks resident.  (Measured, profiles/r03_diag_stream_concurrency.txt: a 128-CU hog changes nothing --
    # 256 blocks still fit; a 250-CU hog starves every kernel of the other stream until it ends.)
    (got, got_len), dt, _ = generate_on_a_starved_device(eng, lambda: eng.generate(eng.prefix, max_new_tokens=n, suppress_eos=True))      # must not raise; must warn
    try:
        assert eng.get_option("chain_fallbacks") == base + 1, "the starved grid was not noticed"
        assert eng.get_option("chain_resident") == 0 and eng.get_option("fuse_qkv_attn") == 0 and eng.get_option("fuse_oproj_fc1") == 0
        assert torch.equal(got.cpu(), want) and list(got_len) == list(want_len)
        # the engine stays on the five-launch chain (no spinning on a device that has shown to be shared): still the same tokens
        again, _ = eng.generate(eng.prefix, max_new_tokens=n, suppress_eos=True)
        assert torch.equal(again.cpu(), want)
        print(f"[fallback] starved generation of {n} tokens took {1e3 * dt:.0f} ms on 32 CUs, incl. one 20 ms bounded sweep and the re-run")
    finally:
        eng.set_option("chain_resident", 1)                          # re-arm for the tests that follow
    assert eng.get_option("chain_resident") == 1
    back, _ = eng.generate(eng.prefix, max_new_tokens=n, suppress_eos=True)
    assert torch.equal(back.cpu(), want)


@pytest.mark.gpu
def test_fc2_inside_the_oproj_fc1_launch_is_bitwise(eng):
    """fuse_fc2: fc2 joins the out_proj + LayerNorm + fc1 launch (relu(fc1) all-gathered inside it, 4096 granules, a quarter per
    wave).  Same arithmetic as the fc2 GEMV launch, so bit-identical logits and tokens -- batch 1 and 2 rows; A/B timing printed."""
    default = eng.get_option("fuse_fc2")
    def run(fuse, prefix, n):
        eng.set_option("fuse_fc2", fuse)
        try:
            toks, _ = eng.generate(prefix, max_new_tokens=n, suppress_eos=True)
            lg = [eng.read_logits(r).clone() for r in range(prefix.shape[0])]
        finally:
            eng.set_option("fuse_fc2", default)
        torch.cuda.synchronize()
        return toks.cpu(), [x.cpu() for x in lg]
    for n in (2, 9, 130):
        t0, g0 = run(0, eng.prefix, n)
        t1, g1 = run(1, eng.prefix, n)
        assert torch.equal(t0, t1), (n, t0.tolist(), t1.tolist())
        assert torch.equal(g0[0].view(torch.int32), g1[0].view(torch.int32)), f"fc2 fused: logits differ after {n} tokens: {float((g0[0] - g1[0]).abs().max()):.3e}"
    two = torch.cat([eng.prefix, eng.prefix.flip(1)])
    t0, g0 = run(0, two, 40)
    t1, g1 = run(1, two, 40)
    assert torch.equal(t0, t1) and all(torch.equal(g0[r].view(torch.int32), g1[r].view(torch.int32)) for r in range(2))
    for L in (300, 3858, eng.cfg.max_seq - 80):
        row = {}
        for fuse in (0, 1, 0, 1):
            eng.set_option("fuse_fc2", fuse)
            eng.profile_decode(L, 2)
            row[fuse] = eng.profile_decode(L, 16)["step_ms_graph"] * 1e3
        eng.set_option("fuse_fc2", default)
        print(f"[fc2 inside oproj+fc1 A/B] kv_len {L:5d}: separate fc2 launch {row[0]:7.1f} us/step | fused {row[1]:7.1f} us/step | ratio {row[1] / row[0]:.3f}")


@pytest.mark.gpu
def test_layer_pair_launch_is_bitwise(eng):
    """fuse_layer: the second half of layer l and the first half of layer l + 1 in one launch (csrc/experimental/layer_fused.hpp; y2 all-gathered
    inside it).  Same device functions as the two launches it replaces: bit-identical logits and tokens, batch 1 and 2 rows."""
    if not eng.get_option("experimental"):
        pytest.skip("the layer-pair launch is not part of the product build (MA_EXPERIMENTAL=1)")
    default = eng.get_option("fuse_layer")
    eng.set_option("fuse_layer", 1)
    available = eng.get_option("fuse_layer") == 1
    eng.set_option("fuse_layer", default)
    if not available:
        pytest.skip("the layer-pair launch exists for the bf16 format only (an experiment, not templated on the 16-bit format)")
    def run(fuse, prefix, n):
        eng.set_option("fuse_layer", fuse)
        try:
            assert eng.get_option("fuse_layer") == fuse
            toks, _ = eng.generate(prefix, max_new_tokens=n, suppress_eos=True)
            lg = [eng.read_logits(r).clone() for r in range(prefix.shape[0])]
        finally:
            eng.set_option("fuse_layer", default)
        torch.cuda.synchronize()
        return toks.cpu(), [x.cpu() for x in lg]
    for n in (2, 9, 130):
        t0, g0 = run(0, eng.prefix, n)
        t1, g1 = run(1, eng.prefix, n)
        assert torch.equal(t0, t1), (n, t0.tolist(), t1.tolist())
        assert torch.equal(g0[0].view(torch.int32), g1[0].view(torch.int32)), f"layer pair: logits differ after {n} tokens: {float((g0[0] - g1[0]).abs().max()):.3e}"
    two = torch.cat([eng.prefix, eng.prefix.flip(1)])
    t0, g0 = run(0, two, 40)
    t1, g1 = run(1, two, 40)
    assert torch.equal(t0, t1) and all(torch.equal(g0[r].view(torch.int32), g1[r].view(torch.int32)) for r in range(2))
    for L in (300, 3858, eng.cfg.max_seq - 80):
        row = {}
        for fuse in (0, 1, 0, 1):
            eng.set_option("fuse_layer", fuse)
            eng.profile_decode(L, 2)
            row[fuse] = eng.profile_decode(L, 16)["step_ms_graph"] * 1e3
        eng.set_option("fuse_layer", default)
        print(f"[layer pair A/B] kv_len {L:5d}: two launches per layer {row[0]:7.1f} us/step | one {row[1]:7.1f} us/step | ratio {row[1] / row[0]:.3f}")
