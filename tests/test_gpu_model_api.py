"""The reference's Python call surface (`meshanything_amd/model.py`) on MI355X: every call a user of the reference makes
(b1 facade, b2 encoder, b3 generate, b4 detokenizer) against the oracle, tiny configuration."""
import types

import numpy as np
import pytest
import torch

from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.config import MAConfig, DTYPE_F32
from conftest import oracle_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from meshanything_amd.model import MeshAnything
    from oracle.meshanything_oracle import Oracle
    cfg = MAConfig.tiny(dtype=DTYPE_F32, max_batch=2)
    args = types.SimpleNamespace(llm="facebook/opt-350m", codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
                                 n_max_triangles=cfg.n_max_faces, ma_config=cfg)
    sd = synthetic_state_dict(cfg, include_unused=True)         # embed_tokens / shape_projection / geo_decoder.* are in the checkpoint too
    model = MeshAnything(args)
    with pytest.raises(Exception):                                  # strict=True: an incomplete checkpoint is an error (main.py:104)
        model.load_state_dict({k: v for k, v in list(sd.items())[:-3]}, strict=True)
    model2 = MeshAnything(args)
    res = model2.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert list(res.missing_keys) == [] and list(res.unexpected_keys) == []
    yield types.SimpleNamespace(cfg=cfg, model=model2, oracle=Oracle(cfg, sd, "fp32", device=oracle_device()))
    model.engine.close()
    model2.engine.close()


def _clouds(cfg, seeds):
    from oracle.meshanything_oracle import normalize_pc
    out = []
    for s in seeds:
        g = torch.Generator().manual_seed(s)
        d = torch.randn(cfg.n_points, 3, generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
        r = 0.3 + 0.7 * torch.rand(cfg.n_points, 1, generator=g)
        out.append(normalize_pc(torch.cat([d * r, d], dim=-1).numpy().astype(np.float32)))
    return torch.from_numpy(np.stack(out))


def test_facade_attributes(env):
    m = env.model
    assert (m.bos_token_id, m.eos_token_id, m.pad_token_id) == (0, 1, 2)
    assert m.cond_length == env.cfg.cond_length and m.face_per_token == 9
    assert m.max_length == env.cfg.n_max_faces * 9 + 2 + env.cfg.cond_length
    assert m.tokenizer.pad_id == -1 and not m.training


def test_each_reference_call(env):
    m, o, cfg = env.model, env.oracle, env.cfg
    x = _clouds(cfg, [21, 22])
    pf = m.point_encoder.encode_latents(x.cuda())                                   # b2
    ref_pf = o.encode_latents(x)
    assert pf.shape == (2, cfg.cond_length, cfg.enc_width)
    assert float((pf.cpu() - ref_pf).abs().max()) < 1e-5
    sl = m.point_encoder.to_shape_latents(pf[:, 1:])
    assert float((sl.cpu() - o.to_shape_latents(ref_pf[:, 1:])).abs().max()) < 2e-5
    prefix = m.process_point_feature(pf)
    ref_prefix = o.process_point_feature(ref_pf)
    assert float((prefix.cpu() - ref_prefix).abs().max()) < 5e-5
    results = m.transformer.generate(inputs_embeds=prefix, max_new_tokens=m.max_length - m.cond_length, num_beams=1,
                                     bos_token_id=0, eos_token_id=1, pad_token_id=2)     # b3, exactly the reference's call
    assert results.dtype == torch.int64 and results.shape[0] == 2 and results.shape[1] <= cfg.max_new_tokens
    ref_tokens = o.generate(ref_prefix)
    from oracle.meshanything_oracle import verify_greedy_stream
    for b in range(2):
        row = results[b].cpu()
        n = int((row == 1).nonzero()[0]) + 1 if (row == 1).any() else row.numel()
        assert verify_greedy_stream(o, ref_prefix[b:b + 1], row[:n], 2e-4)["hard"] == []
    with pytest.raises(NotImplementedError):
        m.transformer.generate(inputs_embeds=prefix, num_beams=4)
    # meshanything.py:163-172 on the host, then get_codes + tokenizer (b4)
    ids = o.postprocess_tokens(results.cpu())
    codes = m.get_codes(ids.cuda())
    ref_codes = o.get_codes(ids)
    assert codes.shape == (2, cfg.n_max_faces * 3, cfg.codebook_dim)
    assert float((codes.cpu() - ref_codes).abs().max()) < 1e-6
    coords = m.tokenizer(ids.cuda(), codes, point_feature=pf)
    ref_coords = o.detokenize(ids, ref_codes, ref_pf)
    assert coords.shape == (2, cfg.n_max_faces, 3, 3) and coords.dtype == torch.float32
    assert torch.equal(torch.isnan(coords.cpu()), torch.isnan(ref_coords))
    assert int((torch.nan_to_num(coords.cpu(), nan=9.0) != torch.nan_to_num(ref_coords, nan=9.0)).sum()) <= 2
    # input_embeds is READ, as in the reference (meshanything.py:53-55): other embeddings -> the oracle's answer for those
    other = torch.roll(ref_codes, shifts=1, dims=1) * 0.5
    coords2 = m.tokenizer(ids.cuda(), other.cuda(), point_feature=pf)
    ref2 = o.detokenize(ids, other, ref_pf)
    assert int((torch.nan_to_num(coords2.cpu(), nan=9.0) != torch.nan_to_num(ref2, nan=9.0)).sum()) <= 2
    assert not torch.equal(torch.nan_to_num(coords2.cpu(), nan=9.0), torch.nan_to_num(coords.cpu(), nan=9.0))
    assert torch.equal(torch.nan_to_num(m.tokenizer(ids.cuda(), None, point_feature=pf), nan=9.0), torch.nan_to_num(coords, nan=9.0))
    del ref_tokens


def test_forward_is_the_composition(env):
    m, cfg = env.model, env.cfg
    x = _clouds(cfg, [23]).cuda()
    out = m(x, sampling=False)                                                       # b1
    assert out.shape == (1, cfg.n_max_faces, 3, 3) and out.dtype == torch.float32
    pf = m.point_encoder.encode_latents(x)
    toks = m.transformer.generate(inputs_embeds=m.process_point_feature(pf))
    ids = m.engine.postprocess_tokens(toks)
    again = m.tokenizer(ids, m.get_codes(ids), point_feature=pf)
    assert torch.equal(torch.nan_to_num(out, nan=9.0), torch.nan_to_num(again, nan=9.0))
    s1 = m.forward_detailed(x, sampling=True, seed=5)["tokens"]
    s2 = m.forward_detailed(x, sampling=True, seed=5)["tokens"]
    assert torch.equal(s1, s2)                                                       # the in-kernel uniform stream is a function of (seed, row, t)


def test_batch_limit_is_enforced(env):
    from meshanything_amd._lib import MAError
    with pytest.raises(MAError):
        env.model(_clouds(env.cfg, [1, 2, 3]).cuda())                                # max_batch = 2


def test_cli_writes_obj_files(tmp_path):
    """`python main.py --input_path mouse.npy --input_type pc_normal ...` end to end (350M shape, seeded synthetic checkpoint,
    8-face cap so that it takes seconds): one OBJ per input, faces index existing vertices."""
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(repo, "tests", "golden", "dataset.npz"))
    src = tmp_path / "mouse.npy"
    np.save(src, g["mouse_raw"])
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(repo, "main.py"), "--input_path", str(src), "--input_type", "pc_normal", "--out_dir", str(out),
                        "--synthetic_weights", "--n_max_triangles", "8", "--seed", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    objs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("_gen.obj")]
    assert len(objs) == 1 and os.path.basename(objs[0]) == "mouse_gen.obj"
    lines = open(objs[0]).read().splitlines()
    nv = sum(l.startswith("v ") for l in lines)
    faces = [[int(t) for t in l.split()[1:]] for l in lines if l.startswith("f ")]
    assert all(1 <= i <= nv for f in faces for i in f)
    assert "Generation Start!!!" in r.stdout and "Over!!" in r.stdout
