"""The released checkpoint is a safetensors file (named .pth, main.py:95-104).  `load_safetensors_items` must feed the engine's
packer the same bytes as in-memory tensors do, in fp32, fp16 and bf16 storage, with either BERT key naming."""
import numpy as np
import pytest
import torch

from meshanything_amd import dp
from meshanything_amd.checkpoint import bf16_round, load_safetensors_items, state_dict_spec, synthetic_state_dict
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F32

safetensors_torch = pytest.importorskip("safetensors.torch")


@pytest.mark.parametrize("policy", [DTYPE_BF16, DTYPE_F32], ids=["bf16", "fp32"])
def test_safetensors_file_packs_like_memory(tmp_path, policy):
    cfg = MAConfig.tiny(dtype=policy)
    sd = synthetic_state_dict(cfg, include_unused=True)             # the file also holds embed_tokens / shape_projection / geo_decoder.*
    ref = dp.pack_host_arena(cfg, sd.items())
    path = tmp_path / "MeshAnything_350m.pth"                       # a safetensors file despite the suffix, like the released one
    safetensors_torch.save_file({k: torch.from_numpy(v) for k, v in sd.items()}, str(path))
    got = dp.pack_host_arena(cfg, load_safetensors_items(str(path)))
    assert np.array_equal(got, ref)
    keys = [k for k, _ in load_safetensors_items(str(path))]
    assert set(keys) == set(state_dict_spec(cfg, include_unused=True))


def test_bf16_stored_safetensors_file(tmp_path):
    """A checkpoint saved in bfloat16 (what `model.to(torch.bfloat16).save_pretrained` writes) loads from disk: numpy has no
    bf16, so the file is read through torch and the packer gets the 16-bit payload."""
    cfg = MAConfig.tiny(dtype=DTYPE_BF16)
    sd = synthetic_state_dict(cfg)
    as_bf16 = {k: torch.from_numpy(v).to(torch.bfloat16) for k, v in sd.items()}
    path = tmp_path / "bf16.safetensors"
    safetensors_torch.save_file(as_bf16, str(path))
    items = list(load_safetensors_items(str(path)))
    assert all(t.dtype == torch.bfloat16 for _, t in items)
    got = dp.pack_host_arena(cfg, items)
    ref = dp.pack_host_arena(cfg, ((k, bf16_round(v)) for k, v in sd.items()))   # same values, stored as fp32
    assert np.array_equal(got, ref)


def test_half_precision_checkpoints_and_fused_bert_names(tmp_path):
    cfg = MAConfig.tiny(dtype=DTYPE_BF16, enc_exact=0)      # (with the exact encoder its matrices stay fp32 in the arena: a bf16-stored file would differ there)
    sd = synthetic_state_dict(cfg)
    # a bf16-stored checkpoint packs to the same arena as its fp32 original whenever the policy rounds to bf16 anyway;
    # 1-D tensors (biases, LayerNorm, tables) stay fp32 in the arena, so only matrices are stored in bf16 here
    mixed = {k: (torch.from_numpy(v).to(torch.bfloat16) if v.ndim == 2 and not k.endswith(("embed_positions.weight", "token_embed_positions.weight",
                                                                                             "cond_embed.weight", "extra_embeds.weight", "pos_embedding.weight",
                                                                                             "point_pe.weight", "encoder.query")) and "quantize_codebooks" not in k
                 else torch.from_numpy(v)) for k, v in sd.items()}
    a = dp.pack_host_arena(cfg, sd.items())
    b = dp.pack_host_arena(cfg, mixed.items())
    assert np.array_equal(a, b)
    fp16 = {k: torch.from_numpy(v).to(torch.float16) for k, v in sd.items()}
    c = dp.pack_host_arena(cfg, fp16.items())                       # accepted; values are those of the fp16 tensors
    assert c.shape == a.shape and not np.array_equal(a, c)
    # optimum BetterTransformer names for the detokenizer layers pack to the same arena as the vanilla HF names
    fused = synthetic_state_dict(cfg, bert_fused=True)
    from oracle.meshanything_oracle import Oracle                   # the oracle's converter is the independent statement of the mapping
    o = Oracle(cfg, fused, "fp32")
    vanilla = {k: v for k, v in o.sd.items()}
    f = dp.pack_host_arena(cfg, fused.items())
    v = dp.pack_host_arena(cfg, ((k, (t.numpy() if hasattr(t, "numpy") else t)) for k, t in vanilla.items() if k in state_dict_spec(cfg)))
    assert np.array_equal(f, v)
    assert np.array_equal(bf16_round(np.float32([1.0, 1.00390625, 3.14159])), np.float32([1.0, 1.0, 3.140625]))


def test_fp16_policy_arena_is_numpy_half_rounding():
    """MA_DTYPE_F16: the host packer's fp32 -> half conversion (round to nearest even, subnormals, overflow to inf) against numpy's, bit for
    bit, on every matrix entry of the arena plus the edge values; an fp16-STORED checkpoint (what the reference's fp16 run would save) packs
    to the same bytes."""
    import ctypes as C
    from meshanything_amd import _lib
    from meshanything_amd.config import DTYPE_F16
    lib = _lib.load()
    cfg = MAConfig.tiny(dtype=DTYPE_F16, enc_exact=0)
    sd = synthetic_state_dict(cfg)
    k = "transformer.lm_head.weight"
    edge = np.array([0.0, -0.0, 1e-8, 5.9604645e-8, 2.98e-8, 6.1e-5, 6.0975552e-5, 65504.0, 65519.9, 65520.0, 1e6, -7e4, 1.0009766, 1.00048828125,
                     1.00146484375, 3.1415927, -2.7182817, np.float32(2.0 ** -24), np.float32(2.0 ** -25), np.float32(1.5 * 2.0 ** -25)], dtype=np.float32)
    w = sd[k].copy()
    w.reshape(-1)[:edge.size] = edge
    sd2 = dict(sd)
    sd2[k] = w
    arena = dp.pack_host_arena(cfg, sd2.items())
    c = cfg.to_c()
    name = C.create_string_buffer(256)
    off, sz = C.c_int64(), C.c_int64()
    dt, rows, cols = C.c_int32(), C.c_int32(), C.c_int32()
    checked = 0
    for i in range(lib.ma_arena_num_entries(C.byref(c))):
        assert lib.ma_arena_entry(C.byref(c), i, name, 256, C.byref(off), C.byref(sz), C.byref(dt), C.byref(rows), C.byref(cols)) == 0
        key = name.value.decode()
        if dt.value != 2 or key not in sd2 or cols.value != sd2[key].shape[-1] or rows.value != sd2[key].shape[0]:
            continue                                                  # fp32 entries; fused q/k/v and padded matrices (covered by the GPU suite)
        got = arena[off.value:off.value + sz.value].view(np.uint16).reshape(rows.value, cols.value)
        with np.errstate(over="ignore"):
            want = sd2[key].astype(np.float16).view(np.uint16)
        assert np.array_equal(got, want), key
        checked += 1
    assert checked >= 10
    with np.errstate(over="ignore"):
        stored = {kk: (v.astype(np.float16) if v.ndim == 2 and kk.endswith(("fc1.weight", "fc2.weight", "lm_head.weight")) else v) for kk, v in sd2.items()}
    assert np.array_equal(dp.pack_host_arena(cfg, stored.items()), arena)
