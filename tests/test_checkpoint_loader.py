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
