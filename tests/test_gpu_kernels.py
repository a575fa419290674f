"""Kernel-level parity (MI355X): each HIP kernel, called through the C ABI (ma_op_*), against a plain PyTorch fp32 / fp64
reference of the same op with the same rounding points.

Inputs are drawn and references computed ON THE GPU with stock torch-ROCm ops (seeded device generator; fp64 matmuls / softmax):
an implementation independent of the HIP library, and nothing here waits for the GPU box's host cores -- on a shared host the
CPU references of this file alone took 250 s of a driver-style run (profiles/r03_gpu_suite_host_cpu_v0.txt)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _gen(seed):
    return torch.Generator(device="cuda").manual_seed(int(seed))


@pytest.fixture(autouse=True)
def _tensors_live_on_the_gpu():
    """Every factory call of a test (randn, linspace, arange, full, ...) creates its tensor on the GPU -- as a context that is popped
    again when the test ends (no process-wide default device is left behind for the other test modules)."""
    torch.backends.cuda.matmul.allow_tf32 = False
    with torch.device("cuda"):
        yield


@pytest.fixture(scope="module")
def lib():
    from meshanything_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return _lib.load()


class _H16:
    """One of the engine's two 16-bit formats (MA_DTYPE_BF16 = 1 | MA_DTYPE_F16 = 2): torch dtype, rounding, the C ABI's code."""
    def __init__(self, name):
        self.name, self.tdt, self.code = name, (torch.bfloat16 if name == "bf16" else torch.float16), (1 if name == "bf16" else 2)

    def rnd(self, x):
        return x.to(self.tdt).to(torch.float32)


@pytest.fixture(params=["bf16", "fp16"])
def h16(request, lib):
    """Runs the test once per 16-bit format: the kernel-level entry points without a dtype argument follow ma_op_set_half_dtype."""
    h = _H16(request.param)
    assert lib.ma_op_set_half_dtype(h.code) == 0
    yield h
    lib.ma_op_set_half_dtype(1)


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk(lib, rc):
    from meshanything_amd import _lib
    _lib.check(rc, None)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _relerr(got, ref):
    got, ref = got.double(), ref.double()
    return float((got - ref).abs().max() / max(1e-6, float(ref.abs().max())))


@pytest.mark.parametrize("wdtype", [0, 1, 2], ids=["f32", "bf16", "fp16"])
@pytest.mark.parametrize("N,K", [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096), (8195, 1024), (128, 128), (256, 128), (67, 256), (128, 2048)])
@pytest.mark.parametrize("variant", ["plain", "ln_relu_res"])
def test_gemv(lib, wdtype, N, K, variant):
    h = _H16("fp16" if wdtype == 2 else "bf16")
    g = _gen(N * 7 + K + wdtype)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    x = torch.randn(K, generator=g) * 1.5 + 0.3
    bias = torch.randn(N, generator=g) * 0.1
    ln = variant == "ln_relu_res"
    lg = 1 + 0.1 * torch.randn(K, generator=g)
    lb = 0.05 * torch.randn(K, generator=g)
    res = torch.randn(N, generator=g)
    # reference
    xr = torch.nn.functional.layer_norm(x, (K,), lg, lb, 1e-5) if ln else x
    Wr = h.rnd(W) if wdtype else W
    xin = h.rnd(xr) if wdtype else xr
    ref = (Wr.double() @ xin.double()).float() + bias
    if ln:
        ref = torch.relu(ref) + res
    dev = "cuda"
    Wd = (W.to(h.tdt) if wdtype else W).to(dev).contiguous()
    y = torch.full((N,), float("nan"), device=dev)
    xn = torch.full((K,), float("nan"), device=dev)
    xd, bd, lgd, lbd, rd = x.to(dev), bias.to(dev), lg.to(dev), lb.to(dev), res.to(dev)
    _chk(lib, lib.ma_op_gemv(wdtype, _p(Wd), _p(bd), _p(xd), _p(lgd) if ln else None, _p(lbd) if ln else None, 1e-5,
                             _p(rd) if ln else None, _p(y), _p(xn) if ln else None, N, K, 1 if ln else 0, _stream()))
    torch.cuda.synchronize()
    assert not torch.isnan(y).any()
    # (16-bit formats behind the LayerNorm prologue: an element that sits on a rounding boundary may round the other way than in the
    #  reference's fp32 LayerNorm -- one ulp of one input element)
    assert _relerr(y, ref) < (5e-5 if (ln and wdtype) else 2e-5), _relerr(y, ref)
    if ln:
        assert _relerr(xn, xr) < 1e-5


@pytest.mark.parametrize("sigmas", [0, 30, 100])
def test_layernorm_prologue_with_an_outlier_in_dim_0(lib, sigmas):
    """The one-pass LayerNorm prologue shifts its statistics by element 0 of the row (csrc/common.hpp).  An outlier THERE is its worst
    case: the error it costs is measured here against fp64 and must stay below the stated budget."""
    N, K = 1024, 1024
    g = _gen(77 + sigmas)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    x = torch.randn(K, generator=g) * 0.8 + 0.4
    if sigmas:
        x[0] = 0.4 + 0.8 * sigmas
    lg, lb = 1 + 0.1 * torch.randn(K, generator=g), 0.05 * torch.randn(K, generator=g)
    xr = torch.nn.functional.layer_norm(x.double(), (K,), lg.double(), lb.double(), 1e-5)
    y = torch.full((N,), float("nan")); xn = torch.full((K,), float("nan"))
    bias = torch.zeros(N)
    _chk(lib, lib.ma_op_gemv(0, _p(W), _p(bias), _p(x), _p(lg), _p(lb), 1e-5, None, _p(y), _p(xn), N, K, 0, _stream()))
    torch.cuda.synchronize()
    err = float((xn.double() - xr).abs().max() / xr.abs().max())
    budget = {0: 2e-6, 30: 3e-4, 100: 3e-3}[sigmas]
    print(f"[one-pass LayerNorm] dim-0 outlier of {sigmas} sigma: max rel error {err:.2e} (budget {budget:.0e})")
    assert err < budget, err


@pytest.mark.parametrize("wdtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("impl", [0, 1], ids=["mfma", "valu"])
@pytest.mark.parametrize("M,N,K,act,use_res", [(257, 768, 768, 0, True), (4096, 768, 64, 0, False), (257, 2304, 768, 0, False),
                                               (1057, 3072, 768, 2, False), (17, 128, 128, 1, True), (1, 1024, 768, 0, False),
                                               (100, 96, 32, 0, False), (256, 64, 3072, 0, True)])
def test_gemm(lib, wdtype, impl, M, N, K, act, use_res):
    g = _gen(M + 3 * N + 5 * K + wdtype)
    # asymmetric data so that a transposed fragment layout cannot pass
    A = torch.randn(M, K, generator=g) + torch.linspace(-1, 1, K)[None, :] * 0.5
    W = torch.randn(N, K, generator=g) / math.sqrt(K) + torch.linspace(0, 1, N)[:, None] * 0.02
    bias = torch.randn(N, generator=g) * 0.1
    R = torch.randn(M, N, generator=g)
    Ar, Wr = (_bf(A), _bf(W)) if wdtype == 1 else (A, W)
    ref = (Ar.double() @ Wr.double().t()).float() + bias
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    if use_res:
        ref = ref + R
    dev = "cuda"
    Ad = A.to(dev).contiguous()
    Wd = (W.to(torch.bfloat16) if wdtype == 1 else W).to(dev).contiguous()
    Cd = torch.full((M, N), float("nan"), device=dev)
    bd, Rd = bias.to(dev), R.to(dev).contiguous()
    _chk(lib, lib.ma_op_gemm(wdtype, impl, _p(Ad), K, _p(Wd), _p(bd), _p(Rd) if use_res else None, N, _p(Cd), N, M, N, K, act, _stream()))
    torch.cuda.synchronize()
    assert not torch.isnan(Cd).any()
    assert _relerr(Cd, ref) < 3e-5, _relerr(Cd, ref)


def test_layernorm(lib):
    g = _gen(3)
    for rows, D, eps in ((257, 768, 1e-5), (1057, 768, 1e-12), (5, 1024, 1e-5), (17, 128, 1e-5)):
        x = torch.randn(rows, D, generator=g) * 3 + 1
        gm, bt = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
        ref = torch.nn.functional.layer_norm(x, (D,), gm, bt, eps)
        xd, y, gd, bd = x.cuda(), torch.empty(rows, D, device="cuda"), gm.cuda(), bt.cuda()
        _chk(lib, lib.ma_op_layernorm(_p(xd), D, _p(gd), _p(bd), eps, _p(y), D, rows, D, _stream()))
        torch.cuda.synchronize()
        assert float((y - ref).abs().max()) < 2e-5


def _attn_ref(q, k, v, scale, causal_offset, rnd, round_p=False):
    # q (Sq,H,64) k,v (Sk,H,64); rnd: q,k,v rounded to 16 bits; round_p: the probabilities that multiply V rounded to 16 bits
    # (True = bf16, or the rounding function of the 16-bit format)
    rq = rnd if callable(rnd) else _bf
    rp = round_p if callable(round_p) else _bf
    if rnd:
        q, k, v = rq(q), rq(k), rq(v)
    w = torch.einsum("qhd,khd->hqk", q.double(), k.double()) * scale
    if causal_offset >= 0:
        Sq, Sk = q.shape[0], k.shape[0]
        mask = torch.arange(Sk)[None, :] > (torch.arange(Sq)[:, None] + causal_offset)
        w = w.masked_fill(mask[None], float("-inf"))
    if round_p:
        pe = torch.exp(w - w.max(dim=-1, keepdim=True).values)
        o = torch.einsum("hqk,khd->qhd", rp(pe.float()).double(), v.double()) / pe.sum(dim=-1).transpose(0, 1)[..., None]
        return o.float().reshape(q.shape[0], -1)
    p = torch.softmax(w, dim=-1)
    return torch.einsum("hqk,khd->qhd", p, v.double()).float().reshape(q.shape[0], -1)


# rnd 0: fp32 policy (exact VALU kernel); 1: bf16 policy = MFMA kernel (q, k, v and P rounded to bf16, fp32 accumulate);
# 2: the VALU kernel on bf16-rounded q, k, v with fp32 probabilities
@pytest.mark.parametrize("rnd", [0, 1, 2], ids=["f32", "bf16mfma", "bf16in"])
@pytest.mark.parametrize("Sq,Sk,H,layout,causal", [(257, 4096, 12, "cross", -1), (257, 257, 12, "interleaved", -1), (257, 257, 16, "std", 0),
                                                    (1057, 1057, 12, "std", -1), (17, 17, 2, "std", 0), (70, 130, 2, "std", 60)])
def test_attention(lib, rnd, Sq, Sk, H, layout, causal):
    g = _gen(Sq + Sk + H)
    q = torch.randn(Sq, H, 64, generator=g)
    k = torch.randn(Sk, H, 64, generator=g)
    v = torch.randn(Sk, H, 64, generator=g)
    ref = _attn_ref(q, k, v, 0.125, causal, rnd != 0, round_p=(rnd == 1))
    dev = "cuda"
    if layout == "std":          # q | k | v blocks of H*64 (OPT / BERT fused projection)
        assert Sq == Sk or True
        Qb = q.reshape(Sq, H * 64).to(dev).contiguous()
        Kb = k.reshape(Sk, H * 64).to(dev).contiguous()
        Vb = v.reshape(Sk, H * 64).to(dev).contiguous()
        args = (Qb, H * 64, 64, Kb, H * 64, 64, Vb, H * 64, 64)
        ptrs = (_p(Qb), H * 64, 64, _p(Kb), H * 64, 64, _p(Vb), H * 64, 64)
    elif layout == "interleaved":  # per head [q|k|v] (transformer_blocks.py:61-62)
        buf = torch.cat([q, k, v], dim=-1).reshape(Sq, H * 192).to(dev).contiguous()
        base = buf.data_ptr()
        ptrs = (C.c_void_p(base), H * 192, 192, C.c_void_p(base + 64 * 4), H * 192, 192, C.c_void_p(base + 128 * 4), H * 192, 192)
        args = (buf,)
    else:                        # cross: q (Sq, H*64); kv per head [k|v] (transformer_blocks.py:172-174)
        Qb = q.reshape(Sq, H * 64).to(dev).contiguous()
        kv = torch.cat([k, v], dim=-1).reshape(Sk, H * 128).to(dev).contiguous()
        base = kv.data_ptr()
        ptrs = (_p(Qb), H * 64, 64, C.c_void_p(base), H * 128, 128, C.c_void_p(base + 64 * 4), H * 128, 128)
        args = (Qb, kv)
    O = torch.full((Sq, H * 64), float("nan"), device=dev)
    _chk(lib, lib.ma_op_attention(*ptrs, _p(O), H * 64, Sq, Sk, H, 0.125, causal, rnd, _stream()))
    torch.cuda.synchronize()
    del args
    assert not torch.isnan(O).any()
    err = float((O - ref).abs().max())
    # MFMA kernel: P is rounded relative to the RUNNING row maximum (online softmax), the reference relative to the final
    # one, so individual p differ by one bf16 ulp (2^-9 relative); the weighted average over keys stays well below that
    assert err < (2e-3 if rnd == 1 else 2e-5), err


@pytest.mark.parametrize("Sq,Sk,H,layout,causal", [(257, 4096, 12, "cross", -1), (257, 257, 12, "interleaved", -1), (257, 257, 16, "std", 0),
                                                    (1057, 1057, 12, "std", -1), (17, 17, 2, "std", 0), (70, 130, 2, "std", 60), (1, 64, 1, "std", -1),
                                                    (96, 65, 3, "interleaved", -1), (128, 128, 2, "std", 0), (300, 1000, 2, "cross", -1)])
def test_attention_bf16_packed_vt(lib, h16, Sq, Sk, H, layout, causal):
    """The engine's dense attention of the bf16 policy (csrc/attn2.hpp: V^T packing + swapped-operand 32x32x16 MFMA kernel) on bf16
    tensors in the three layouts the engine uses, ragged and causal cases, against fp64 softmax on the same bf16 inputs with the
    probabilities that multiply V rounded to bf16 (the policy's rounding points); the output itself is bf16 (half an ulp of O(1))."""
    g = _gen(3 * Sq + Sk + H)
    q = h16.rnd(torch.randn(Sq, H, 64, generator=g))
    k = h16.rnd(torch.randn(Sk, H, 64, generator=g) + torch.linspace(-0.3, 0.3, 64)[None, None, :])      # asymmetric: a swapped operand cannot pass
    v = h16.rnd(torch.randn(Sk, H, 64, generator=g) + torch.linspace(0.5, -0.5, 64)[None, None, :])
    ref = _attn_ref(q, k, v, 0.125, causal, False, round_p=h16.rnd)
    b16 = lambda t: t.to(h16.tdt).contiguous()
    if layout == "std":
        Qb, Kb, Vb = b16(q.reshape(Sq, H * 64)), b16(k.reshape(Sk, H * 64)), b16(v.reshape(Sk, H * 64))
        ptrs = (_p(Qb), H * 64, 64, _p(Kb), H * 64, 64, _p(Vb), H * 64, 64)
        keep = (Qb, Kb, Vb)
    elif layout == "interleaved":          # per head [q|k|v] (transformer_blocks.py:61-62); needs Sq == Sk rows in one buffer
        n = max(Sq, Sk)
        buf = torch.zeros(n, H, 192)
        buf[:Sq, :, :64] = q; buf[:Sk, :, 64:128] = k; buf[:Sk, :, 128:] = v
        buf = b16(buf.reshape(n, H * 192))
        base = buf.data_ptr()
        ptrs = (C.c_void_p(base), H * 192, 192, C.c_void_p(base + 64 * 2), H * 192, 192, C.c_void_p(base + 128 * 2), H * 192, 192)
        keep = (buf,)
    else:                                  # cross: q (Sq, H*64); kv per head [k|v] (transformer_blocks.py:172-174)
        Qb = b16(q.reshape(Sq, H * 64))
        kv = b16(torch.cat([k, v], dim=-1).reshape(Sk, H * 128))
        base = kv.data_ptr()
        ptrs = (_p(Qb), H * 64, 64, C.c_void_p(base), H * 128, 128, C.c_void_p(base + 64 * 2), H * 128, 128)
        keep = (Qb, kv)
    O = torch.full((Sq, H * 64), float("nan"), dtype=h16.tdt)
    _chk(lib, lib.ma_op_attention(*ptrs, _p(O), H * 64, Sq, Sk, H, 0.125, causal, 4, _stream()))
    torch.cuda.synchronize()
    del keep
    assert not torch.isnan(O.float()).any()
    err = float((O.float() - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    # P is rounded relative to the RUNNING row maximum (online softmax), the reference relative to the final one: single probabilities
    # differ by one bf16 ulp; plus the bf16 rounding of the output (2^-9 relative)
    assert err < 6e-3 * scale, err
    assert float((O.float() - ref).abs().mean()) < 1.5e-3 * scale


@pytest.mark.parametrize("kvdtype", [0, 1, 2], ids=["f32", "bf16", "fp16"])
@pytest.mark.parametrize("length,max_seq,H", [(1, 7459, 16), (5, 7459, 16), (128, 7459, 16), (129, 7459, 16), (257, 7459, 16), (300, 99, 2), (1000, 7459, 16),
                                               (7459, 7459, 16), (14659, 14659, 16)])
def test_decode_attention(lib, kvdtype, length, max_seq, H):
    """Split-KV decode attention (16 equal chunks per head) + the partial merge; repeated launches reuse the same
    workspace and the result must be bit-stable from launch to launch."""
    max_seq = max(max_seq, length)
    g = _gen(length + H)
    q = torch.randn(H * 64, generator=g)
    k = torch.randn(H, max_seq, 64, generator=g)
    v = torch.randn(H, max_seq, 64, generator=g)
    h = _H16("fp16" if kvdtype == 2 else "bf16")
    rnd = h.rnd if kvdtype else False
    kk, vv = k[:, :length].permute(1, 0, 2), v[:, :length].permute(1, 0, 2)
    ref = _attn_ref(q.reshape(1, H, 64), kk, vv, 0.125, -1, rnd)[0]
    dev = "cuda"
    kd = (k.to(h.tdt) if rnd else k).to(dev).contiguous()
    vd = (v.to(h.tdt) if rnd else v).to(dev).contiguous()
    nbytes = lib.ma_decode_attention_workspace_bytes(H)
    assert nbytes == H * 16 * 66 * 4
    ws = torch.empty(nbytes // 4, device=dev)
    qd = q.to(dev)
    outs = []
    for it in range(3):
        out = torch.full((H * 64,), float("nan"), device=dev)
        _chk(lib, lib.ma_op_decode_attention(kvdtype, _p(qd), _p(kd), _p(vd), H, max_seq, length, _p(out), _p(ws), _stream()))
        torch.cuda.synchronize()
        outs.append(out)
    assert not torch.isnan(outs[0]).any()
    assert float((outs[0] - ref).abs().max()) < 2e-5
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("waves", [4, 8, 16, 82])        # 82: 8 waves, two blocks per (row, head) with the in-launch hand-over
@pytest.mark.parametrize("B,length,H", [(16, 1, 16), (16, 257, 16), (17, 130, 2), (8, 700, 16), (11, 2500, 16), (40, 1000, 16), (64, 1500, 16)])
def test_decode_attention_rows(lib, h16, B, length, H, waves):
    """Final-form batched decode attention (>= 16 rows: one block per (row, head), normalised bf16 output, no merge launch):
    every row against the fp32 softmax reference on the same bf16-rounded q / K / V; bit-stable across launches; rows of the
    cache beyond `length` hold NaN and must not be touched."""
    if waves not in (4, 82) and B * length > 20000 or waves == 82 and B * length > 40000:
        pytest.skip("the large cases run once, with the default block size")
    g = _gen(B * 7 + length)
    max_seq = length + 3
    q = torch.randn(B, H * 64, generator=g)
    k = torch.randn(B, H, max_seq, 64, generator=g)
    v = torch.randn(B, H, max_seq, 64, generator=g)
    k[:, :, length:] = float("nan"); v[:, :, length:] = float("nan")
    kd, vd = k.to(h16.tdt).cuda().contiguous(), v.to(h16.tdt).cuda().contiguous()
    qd = q.cuda()
    outs = []
    for it in range(2):
        out = torch.full((B, H * 64), float("nan"), device="cuda", dtype=h16.tdt)
        _chk(lib, lib.ma_op_decode_attention_rows(_p(qd), _p(kd), _p(vd), H, max_seq, length, B, H * max_seq * 64, 8 if waves == 82 else waves, 2 if waves == 82 else 1, _p(out), _stream()))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and not torch.isnan(outs[0].float()).any()
    qb = h16.rnd(q).reshape(B, H, 64)
    kf, vf = kd.float()[:, :, :length], vd.float()[:, :, :length]
    p = torch.softmax(torch.einsum("bhd,bhsd->bhs", qb.double(), kf.double()) * 0.125, dim=-1)
    ref = torch.einsum("bhs,bhsd->bhd", p, vf.double()).reshape(B, H * 64).float()
    # output is rounded to bf16 (2^-9 relative); values are O(1)
    assert float((outs[0].float() - ref).abs().max()) < 1.5e-2
    assert float((outs[0].float() - h16.rnd(ref)).abs().mean()) < 1e-3


@pytest.mark.parametrize("B,length,waves", [(8, 9000, 82), (8, 14659, 82), (8, 14659, 8), (10, 14659, 8), (12, 14659, 4), (64, 7459, 4)])
def test_decode_attention_rows_deep_cache(lib, h16, B, length, waves):
    """The final-form decode attention at the cache depths of BASELINE configs 3 and 5 (the kernel test stopped at 2 500
    positions): 7 459 positions x 64 rows (config 3's last step), 9 000 and 14 659 positions x 8 rows in the two-block form the engine
    runs at 8 rows (waves 82) and in the one-block forms, 10 / 12 rows at 14 659.  Operands are drawn on the device (the largest case holds
    3.9 GB of K / V); every (row, head) against an fp64 softmax reference on the same rounded q / K / V; bit-stable across launches; cache rows
    beyond `length` hold NaN and must not be touched."""
    H = 16
    max_seq = length + 3
    g = torch.Generator(device="cuda").manual_seed(B * 7 + length)
    q = torch.randn(B, H * 64, generator=g, device="cuda")
    kd = torch.randn(B, H, max_seq, 64, generator=g, device="cuda").to(h16.tdt)
    vd = torch.randn(B, H, max_seq, 64, generator=g, device="cuda").to(h16.tdt)
    kd[:, :, length:] = float("nan"); vd[:, :, length:] = float("nan")
    outs = []
    for it in range(2):
        out = torch.full((B, H * 64), float("nan"), device="cuda", dtype=h16.tdt)
        _chk(lib, lib.ma_op_decode_attention_rows(_p(q), _p(kd), _p(vd), H, max_seq, length, B, H * max_seq * 64, 8 if waves == 82 else waves, 2 if waves == 82 else 1, _p(out), _stream()))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and not torch.isnan(outs[0].float()).any()
    qb = q.to(h16.tdt).double().reshape(B, H, 64)
    worst, mean = 0.0, 0.0
    for b in range(B):                                     # one row at a time: the fp64 copies of a row's K / V are 240 MB
        kf, vf = kd[b, :, :length].double(), vd[b, :, :length].double()
        p = torch.softmax(torch.einsum("hd,hsd->hs", qb[b], kf) * 0.125, dim=-1)
        ref = torch.einsum("hs,hsd->hd", p, vf).reshape(H * 64).float()
        worst = max(worst, float((outs[0][b].float() - ref).abs().max()))
        mean += float((outs[0][b].float() - ref.to(h16.tdt).float()).abs().mean()) / B
    # (a softmax over ~10^4 random scores is nearly flat: outputs are O(0.01-0.1), rounded to 16 bits at the end)
    assert worst < 2e-3 and mean < 1e-4, (worst, mean)


# ---------------------------------------------------------------------------------------------- batched decode step kernels
@pytest.mark.parametrize("B", [1, 4, 8, 13, 16])
@pytest.mark.parametrize("N,parts,act,extras", [(3072, 4, 0, True), (4096, 1, 1, True), (4096, 1, 1, False), (1024, 2, 0, True), (8195, 4, 0, True)])
def test_gemm_dec_ln(lib, h16, B, N, parts, act, extras):
    """Skinny GEMM with the LayerNorm prologue inside (small batches of the batched decode step): activation row = LN(sum of the
    partial buffers + bias + residual) rounded to bf16; against fp64 torch on the same rounding point; the LayerNorm output itself
    (the later residual) within fp32 accuracy; bit-stable across launches."""
    K = 1024
    g = _gen(B + N + parts)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(h16.tdt)
    pin = torch.randn(parts, B, K, generator=g) * 0.7 + 3.0            # a large common offset: the shifted statistics must cope
    pb = torch.randn(K, generator=g) * 0.1 if extras else None
    pr = torch.randn(B, K, generator=g) if extras else None
    lg, lb = 1.0 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    bias = torch.randn(N, generator=g) * 0.1
    x = pin.double().sum(dim=0)
    if extras:
        x = x + pb.double() + pr.double()
    xn = torch.nn.functional.layer_norm(x, (K,), lg.double(), lb.double(), 1e-5)
    dev = "cuda"
    d = lambda t: None if t is None else t.to(dev).contiguous()
    Wd, pind, pbd, prd, lgd, lbd, bd = d(W), d(pin), d(pb), d(pr), d(lg), d(lb), d(bias)
    outs = []
    for it in range(2):
        y = torch.full((B, N), float("nan"), device=dev)
        yb = torch.zeros(B, N, dtype=h16.tdt, device=dev)
        xo = torch.full((B, K), float("nan"), device=dev)
        _chk(lib, lib.ma_op_gemm_dec_ln(_p(Wd), _p(bd), _p(pind), parts, _p(pbd), _p(prd), _p(lgd), _p(lbd), 1e-5, _p(xo), _p(y), _p(yb), N, B, act, _stream()))
        torch.cuda.synchronize()
        outs.append((y, yb, xo))
    y, yb, xo = outs[0]
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    assert float((xo.double() - xn).abs().max()) < 2e-5
    ref = xo.to(h16.tdt).double() @ W.double().t() + bias.double()   # the kernel's own LayerNorm output, rounded where the kernel rounds
    if act == 1:
        ref = torch.relu(ref)
    assert not torch.isnan(y).any()
    assert _relerr(y, ref.float()) < 2e-5, _relerr(y, ref.float())
    assert torch.equal(yb, y.to(h16.tdt))


@pytest.mark.parametrize("B", [4, 16, 17, 40, 64])
@pytest.mark.parametrize("N,K,ksplit,act", [(3072, 1024, 1, 0), (4096, 1024, 1, 1), (1024, 1024, 4, 0), (1024, 4096, 4, 0), (1024, 4096, 1, 0),
                                           (8195, 1024, 1, 0), (384, 128, 1, 1), (128, 512, 4, 0)])
def test_gemm_dec(lib, h16, B, N, K, ksplit, act):
    """Skinny bf16 MFMA GEMM of the batched decode step (gemm_decode.hpp) against fp64 torch: every batch-tile count (1-4),
    whole-K and split-K launches (split launches return raw partial sums, summed here), bias / ReLU / residual epilogue,
    fp32 and bf16 outputs."""
    g = _gen(B + N + 3 * K + ksplit)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K) + torch.linspace(0, 1, N)[:, None] * 0.02).to(h16.tdt)
    X = (torch.randn(B, K, generator=g) + torch.linspace(-1, 1, K)[None, :] * 0.5).to(h16.tdt)
    bias = torch.randn(N, generator=g) * 0.1
    res = torch.randn(B, N, generator=g)
    ref = X.double() @ W.double().t()
    dev = "cuda"
    Wd, Xd = W.to(dev).contiguous(), X.to(dev).contiguous()
    if ksplit > 1:
        y = torch.full((ksplit, B, N), float("nan"), device=dev)
        _chk(lib, lib.ma_op_gemm_dec(_p(Wd), None, _p(Xd), None, _p(y), None, N, K, B, 0, ksplit, _stream()))
        torch.cuda.synchronize()
        assert not torch.isnan(y).any()
        assert _relerr(y.sum(dim=0), ref.float()) < 2e-5
        return
    full = ref + bias.double()
    if act == 1:
        full = torch.relu(full)
    full = (full + res.double()).float()
    bd, rd = bias.to(dev), res.to(dev).contiguous()
    y = torch.full((B, N), float("nan"), device=dev)
    yb = torch.zeros(B, N, dtype=h16.tdt, device=dev)
    _chk(lib, lib.ma_op_gemm_dec(_p(Wd), _p(bd), _p(Xd), _p(rd), _p(y), _p(yb), N, K, B, act, 1, _stream()))
    torch.cuda.synchronize()
    assert not torch.isnan(y).any()
    assert _relerr(y, full) < 2e-5, _relerr(y, full)
    assert torch.equal(yb, y.to(h16.tdt))          # the 16-bit output is the rounded fp32 output


@pytest.mark.parametrize("B", [4, 17, 64])
def test_gemm_dec_qkv_epilogue(lib, h16, B):
    """q rows -> fp32 vector per batch row; k / v rows -> that row's cache planes at `pos` (bf16), nothing else touched."""
    H, heads, max_seq, pos = 1024, 16, 300, 271
    g = _gen(B)
    W = (torch.randn(3 * H, H, generator=g) / math.sqrt(H)).to(h16.tdt)
    X = torch.randn(B, H, generator=g).to(h16.tdt)
    bias = torch.randn(3 * H, generator=g) * 0.1
    ref = (X.double() @ W.double().t() + bias.double()).float()
    dev = "cuda"
    stride = heads * max_seq * 64
    kc = torch.full((B, heads, max_seq, 64), 7.0, dtype=h16.tdt, device=dev)
    vc = torch.full((B, heads, max_seq, 64), -7.0, dtype=h16.tdt, device=dev)
    q = torch.full((B, H), float("nan"), device=dev)
    Wd, Xd, bd = W.to(dev).contiguous(), X.to(dev).contiguous(), bias.to(dev)
    _chk(lib, lib.ma_op_gemm_dec_qkv(_p(Wd), _p(bd), _p(Xd), _p(q), _p(kc), _p(vc), H, max_seq, pos, B, stride, _stream()))
    torch.cuda.synchronize()
    assert _relerr(q, ref[:, :H]) < 2e-5
    kref = ref[:, H:2 * H].reshape(B, heads, 64).to(h16.tdt)
    vref = ref[:, 2 * H:].reshape(B, heads, 64).to(h16.tdt)
    kgot, vgot = kc[:, :, pos], vc[:, :, pos]
    # bf16 rounding of values that differ by fp32 summation order may land one ulp apart
    assert float((kgot.float() - kref.float()).abs().max()) <= 2 ** -6 and float((vgot.float() - vref.float()).abs().max()) <= 2 ** -6
    assert float((kgot.float() - kref.float()).abs().mean()) < 1e-4
    kc[:, :, pos] = 7.0
    vc[:, :, pos] = -7.0
    assert bool((kc == 7.0).all()) and bool((vc == -7.0).all())       # no other position was written


@pytest.mark.parametrize("B", [4, 17, 64])
@pytest.mark.parametrize("pro", [0, 1, 2], ids=["plain", "ln", "attn"])
def test_rows_prologue(lib, h16, B, pro):
    """Per-row prologue of the batched step: split-K partial sum + bias + residual (+ LayerNorm), or the merge of the
    split-KV attention partials; fp32 and bf16 outputs."""
    K, heads = 1024, 16
    g = _gen(B + pro)
    dev = "cuda"
    xb = torch.zeros(B, K, dtype=h16.tdt, device=dev)
    xn = torch.full((B, K), float("nan"), device=dev)
    if pro == 2:
        m = torch.randn(B, heads, 16, generator=g) * 2
        l = torch.rand(B, heads, 16, generator=g) + 0.5
        o = torch.randn(B, heads, 16, 64, generator=g)
        ws = torch.cat([torch.stack([m, l], dim=-1).reshape(B, -1), o.reshape(B, -1)], dim=1).contiguous()      # ML[h][c][2] then O[h][c][64]
        wmax = m.max(dim=-1, keepdim=True).values
        f = torch.exp(m.double() - wmax.double())
        ref = ((o.double() * f[..., None]).sum(dim=2) / (l.double() * f).sum(dim=2)[..., None]).reshape(B, K).float()
        wsd = ws.to(dev)
        _chk(lib, lib.ma_op_rows_prologue(2, None, 1, B, None, None, None, None, 0.0, _p(wsd), heads, _p(xn), _p(xb), K, _stream()))
    else:
        nparts = 4
        parts = torch.randn(nparts, B, K, generator=g)
        bias = torch.randn(K, generator=g) * 0.1
        res = torch.randn(B, K, generator=g) + 3.0                        # a large mean: the one-pass statistics must not care
        lg = 1 + 0.1 * torch.randn(K, generator=g)
        lb = 0.05 * torch.randn(K, generator=g)
        x = parts.double().sum(dim=0) + bias.double() + res.double()
        ref = (torch.nn.functional.layer_norm(x, (K,), lg.double(), lb.double(), 1e-5) if pro == 1 else x).float()
        pd, bd, rd, lgd, lbd = parts.to(dev).contiguous(), bias.to(dev), res.to(dev).contiguous(), lg.to(dev), lb.to(dev)
        _chk(lib, lib.ma_op_rows_prologue(pro, _p(pd), nparts, B, _p(bd), _p(rd), _p(lgd) if pro == 1 else None, _p(lbd) if pro == 1 else None, 1e-5,
                                           None, heads, _p(xn), _p(xb), K, _stream()))
    torch.cuda.synchronize()
    assert not torch.isnan(xn).any()
    assert float((xn - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(xb, xn.to(h16.tdt))


@pytest.mark.parametrize("M,N,K,act,use_res,out", [(4112, 3072, 1024, 0, False, "bf16"), (4112, 1024, 1024, 0, True, "f32"), (4112, 4096, 1024, 1, False, "bf16"),
                                                   (4112, 1024, 4096, 0, True, "f32"), (2056, 768, 768, 2, False, "both"), (65536, 1536, 768, 0, False, "bf16"),
                                                   (257, 2304, 768, 0, False, "bf16"), (130, 200, 192, 0, True, "both"), (1, 1024, 768, 0, False, "f32"),
                                                   (16912, 768, 3072, 0, True, "f32"), (300, 64, 768, 0, False, "f32"), (77, 1152, 96, 0, False, "f32")])
@pytest.mark.parametrize("variant", [0, 6, 12, 13], ids=["syncthreads", "rawbarrier", "tile256x128", "tile256x128s3"])
def test_gemm_bf16_tile(lib, h16, M, N, K, act, use_res, out, variant):
    """The bf16 policy's dense GEMM (gemm_tile.hpp) on its native bf16 operands at the batched dense-phase shapes (B x 257,
    B x 4096, B x 1057 rows), ragged edges, both tile variants; fp32 and bf16 outputs; reports TFLOP/s.  variant: the K-loop's
    barrier form and tile (engine option gemm_variant: 0 = __syncthreads(), 6 = counted vmcnt + raw s_barrier with the per-shape tile choice
    of launch_gemm_tile, the default; 12 / 13 = the 8-wave 256 x 128 tile with two / three LDS stages everywhere)."""
    from meshanything_amd.config import MAConfig, DTYPE_BF16
    from meshanything_amd.engine import Engine
    knob = Engine(MAConfig.tiny(dtype=DTYPE_BF16))                      # gemm_variant is a process-wide knob behind an engine option
    if variant != 6 and (not knob.get_option("experimental") or h16.name == "fp16"):
        knob.close()
        pytest.skip("the A/B tile variants live in libraries built with MA_EXPERIMENTAL=1 (and are exercised in bf16 there)")
    knob.set_option("gemm_variant", variant)
    g = _gen(M + 3 * N + 5 * K)
    A = (torch.randn(M, K, generator=g) + torch.linspace(-1, 1, K)[None, :] * 0.5).to(h16.tdt)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K) + torch.linspace(0, 1, N)[:, None] * 0.02).to(h16.tdt)
    bias = torch.randn(N, generator=g) * 0.1
    R = torch.randn(M, N, generator=g)
    dev = "cuda"
    Ad, Wd, bd, Rd = A.to(dev), W.to(dev), bias.to(dev), R.to(dev)
    ref = Ad.double() @ Wd.double().t() + bd.double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    if use_res:
        ref = ref + Rd.double()
    ref = ref.float()
    C = torch.full((M, N), float("nan"), device=dev) if out in ("f32", "both") else None
    Cb = torch.zeros(M, N, dtype=h16.tdt, device=dev) if out in ("bf16", "both") else None

    def run():
        _chk(lib, lib.ma_op_gemm_bf16(_p(Ad), K, _p(Wd), _p(bd), _p(Rd) if use_res else None, N, _p(C), N, _p(Cb), N, M, N, K, act, _stream()))
    run()
    torch.cuda.synchronize()
    scale = max(1e-6, float(ref.abs().max()))
    if C is not None:
        assert not torch.isnan(C).any()
        assert float((C - ref).abs().max()) / scale < 3e-5
    if Cb is not None:
        assert float((Cb.float() - ref).abs().max()) / scale < 6e-3          # bf16 output: half an ulp of the largest value
        if C is not None:
            assert torch.equal(Cb, C.to(h16.tdt))
    if M * N * K >= 1 << 30:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(3):
            run()
        ev[0].record()
        for _ in range(20):
            run()
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / 20 * 1e3
        print(f"[gemm_tile, variant {variant}] M {M} N {N} K {K} act {act} res {use_res} out {out}: {us:.1f} us = {2.0 * M * N * K / us * 1e-6:.1f} TFLOP/s")
    knob.set_option("gemm_variant", 6)
    knob.close()


@pytest.mark.parametrize("M,N,K,act,use_res,out", [(16448, 1024, 1024, 0, True, "f32"), (16448, 3072, 1024, 0, False, "bf16"), (16448, 4096, 1024, 1, False, "bf16"),
                                                   (16448, 1024, 4096, 0, True, "f32"), (67648, 768, 768, 2, False, "both"), (8192, 8192, 512, 0, False, "bf16"),
                                                   (4097, 4352, 128, 0, True, "both"), (33000, 1152, 768, 0, False, "f32"), (66000, 256, 64, 0, False, "f32"),
          (67648, 2304, 768, 0, False, "bf16"), (16448, 3072, 768, 2, False, "bf16"), (4112, 3072, 1024, 0, False, "bf16")])
def test_gemm_256_tile(lib, h16, M, N, K, act, use_res, out):
    """The 256 x 256 x 64 eight-wave tile with the staged K-loop (csrc/gemm256.hpp) at the batched dense-phase shapes -- M = 64 x 257 (65 tile
    rows: whole rounds on the big tile + the remainder on 128 x 128 tiles), M = 64 x 1057, ragged M and N edges, K = 64 (one K-tile: prologue only)
    and K = 128 -- against fp64 torch; bit-stable across launches (its in-flight LDS-DMA schedule is timing dependent, its result must not be);
    A/B timing against the 128-row tiles it replaces."""
    from meshanything_amd.config import MAConfig, DTYPE_BF16
    from meshanything_amd.engine import Engine
    knob = Engine(MAConfig.tiny(dtype=DTYPE_BF16))                      # gemm256 is a process-wide knob behind an engine option
    g = _gen(M + 3 * N + 5 * K)
    A = (torch.randn(M, K, generator=g) + torch.linspace(-1, 1, K)[None, :] * 0.5).to(h16.tdt)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K) + torch.linspace(0, 1, N)[:, None] * 0.02).to(h16.tdt)
    bias = torch.randn(N, generator=g) * 0.1
    R = torch.randn(M, N, generator=g) if use_res else None
    ref = A.double() @ W.double().t() + bias.double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    if use_res:
        ref = ref + R.double()
    ref = ref.float()
    scale = max(1e-6, float(ref.abs().max()))

    def run(Cf, Cb):
        _chk(lib, lib.ma_op_gemm_bf16(_p(A), K, _p(W), _p(bias), _p(R), N, _p(Cf), N, _p(Cb), N, M, N, K, act, _stream()))
    res = {}
    for mode in (2, 1, 0, 2):                            # 2: + the persistent form where it applies (whole-tile 16-bit output, more than one round)
        knob.set_option("gemm256", mode)
        Cf = torch.full((M, N), float("nan")) if out in ("f32", "both") else None
        Cb = torch.zeros(M, N, dtype=h16.tdt) if out in ("bf16", "both") else None
        run(Cf, Cb)
        torch.cuda.synchronize()
        def where(bad):                                  # (a failure of this test was seen once without its message: say everything)
            idx = bad.nonzero()
            return f"mode {mode}: {int(bad.sum())} elements, rows {int(idx[:, 0].min())}..{int(idx[:, 0].max())}, cols {int(idx[:, 1].min())}..{int(idx[:, 1].max())}, first {idx[0].tolist()}"
        if Cf is not None:
            assert not torch.isnan(Cf).any(), "NaN left in the fp32 output: " + where(torch.isnan(Cf))
            e32 = (Cf - ref).abs() / scale
            assert float(e32.max()) < 3e-5, f"fp32 output off by {float(e32.max()):.3e}: " + where(e32 >= 3e-5)
        if Cb is not None:
            e16 = (Cb.float() - ref).abs() / scale
            assert float(e16.max()) < 6e-3, f"16-bit output off by {float(e16.max()):.3e}: " + where(e16 >= 6e-3)
            if Cf is not None:
                assert torch.equal(Cb, Cf.to(h16.tdt)), "16-bit copy is not the rounded fp32 output: " + where(Cb != Cf.to(h16.tdt))
        if mode == 2 and 2 in res:
            for a_, b_ in zip(res[2], (Cf, Cb)):
                assert a_ is None or torch.equal(a_, b_), "the 256 x 256 kernel is not bit-stable from launch to launch: " + where(a_ != b_)
        res[mode] = (Cf, Cb)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(2):
            run(Cf, Cb)
        ev[0].record()
        for _ in range(10):
            run(Cf, Cb)
        ev[1].record()
        torch.cuda.synchronize()
        res[("us", mode)] = ev[0].elapsed_time(ev[1]) / 10 * 1e3
    # both tile kernels accumulate k-ascending in 32-wide MFMA steps: identical results, not just close ones -- on the rows the big tile
    # computes (a tail of <= 64 rows goes to the skinny GEMM of the batched decode step, whose four waves split K)
    same_rows = M - (M % 256 if M % 256 <= 64 else 0)
    for a_, b_ in zip(res[0], res[1]):
        assert a_ is None or torch.equal(a_[:same_rows], b_[:same_rows]), "256 x 256 and 128-row tiles disagree bitwise: " + str(int((a_[:same_rows] != b_[:same_rows]).sum())) + " elements"
    for a_, b_ in zip(res[2], res[1]):
        assert a_ is None or torch.equal(a_, b_), "the persistent and the one-tile 256 x 256 kernels disagree bitwise: " + str(int((a_ != b_).sum())) + " elements"
    fl = 2.0 * M * N * K
    print(f"[gemm256 {h16.name}] M {M} N {N} K {K} act {act} res {use_res} out {out}: default (persistent where it applies) {res[('us', 2)]:.1f} us = {fl / res[('us', 2)] * 1e-6:.1f} TFLOP/s | "
          f"one tile per workgroup {res[('us', 1)]:.1f} us = {fl / res[('us', 1)] * 1e-6:.1f} | 128-row tiles {res[('us', 0)]:.1f} us = {fl / res[('us', 0)] * 1e-6:.1f} | ratio {res[('us', 0)] / res[('us', 2)]:.2f}x")
    knob.set_option("gemm256", 2)
    knob.close()
