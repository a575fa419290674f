"""The oracle on torch-ROCm against the oracle on the CPU (tiny configuration, both precision policies): the GPU suite runs
its long verifications with `Oracle(device="cuda")` so that they do not depend on the box's host cores; this keeps that choice
honest -- same statements, two devices, results within summation-order noise, identical token decisions."""
import numpy as np
import pytest
import torch

from meshanything_amd.config import MAConfig
from meshanything_amd.checkpoint import synthetic_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("policy", ["fp32", "bf16"])
def test_oracle_cuda_equals_oracle_cpu(policy):
    from oracle.meshanything_oracle import Oracle, normalize_pc, verify_greedy_stream, verify_sampled_stream
    cfg = MAConfig.tiny()
    sd = synthetic_state_dict(cfg)
    cpu, dev = Oracle(cfg, sd, policy), Oracle(cfg, sd, policy, device="cuda")
    g = torch.Generator().manual_seed(21)
    d = torch.randn(2, cfg.n_points, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    cloud = torch.cat([d * (0.3 + 0.7 * torch.rand(2, cfg.n_points, 1, generator=g)), d], -1).numpy().astype(np.float32)
    x = torch.from_numpy(np.stack([normalize_pc(c) for c in cloud]))
    tol = 2e-5 if policy == "fp32" else 2e-2          # bf16: one rounding flip of an activation moves a logit by ~1e-2
    a, b = cpu.forward(x, suppress_eos=True), dev.forward(x, suppress_eos=True)
    assert all(v.device.type == "cpu" for v in b.values()), "public oracle calls hand their results back on the CPU"
    assert float((a["point_feature"] - b["point_feature"]).abs().max()) < tol
    assert float((a["prefix"] - b["prefix"]).abs().max()) < 4 * tol
    for r in range(2):
        # each device's greedy stream is a valid greedy decode under the other device's arithmetic
        for ora, other in ((cpu, b), (dev, a)):
            v = verify_greedy_stream(ora, a["prefix"][r:r + 1], other["tokens"][r], 10 * tol, suppress_eos=True)
            assert v["hard"] == [], v
        la = cpu.teacher_forced_logits(a["prefix"][r:r + 1], a["tokens"][r])
        lb = dev.teacher_forced_logits(a["prefix"][r:r + 1], a["tokens"][r])
        assert float((la - lb).abs().max()) < 10 * tol
    if torch.equal(a["tokens"], b["tokens"]):
        assert torch.equal(a["ids"], b["ids"])
        assert int((torch.nan_to_num(a["coords"], nan=9.0) != torch.nan_to_num(b["coords"], nan=9.0)).sum()) <= 2
    u = np.random.default_rng(4).random((1, cfg.max_new_tokens)).astype(np.float32)
    s = dev.generate(a["prefix"][:1], sampling=True, uniforms=u, suppress_eos=True)
    v = verify_sampled_stream(cpu, a["prefix"][:1], s[0], u[0], tol=10 * tol, suppress_eos=True)
    assert v["hard"] == [], v
