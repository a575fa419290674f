"""The oracle on torch-ROCm against the oracle on the CPU (tiny configuration, both precision policies): the GPU suite runs
its long verifications with `Oracle(device="cuda")` so that they do not depend on the box's host cores; this keeps that choice
honest -- same statements, two devices, results within summation-order noise, identical token decisions.

The CPU side runs in a FRESH interpreter (subprocess, one thread, 120 s limit): inside the long-lived pytest process of a
driver-style run the same tiny-shape CPU calls took 150 s instead of 0.1 s (profiles/r03_gpu_suite_untasksetted_v2.txt; not
reproducible in a fresh process on the same box, profiles/r03_diag_oracle_devices.txt), and nothing in the GPU suite may hang on
the host like that again."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from meshanything_amd.config import MAConfig
from meshanything_amd.checkpoint import synthetic_state_dict

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CPU_SIDE = r"""
import os, sys
for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"): os.environ[k] = "1"
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
torch.set_num_threads(1)
from meshanything_amd.config import MAConfig
from meshanything_amd.checkpoint import synthetic_state_dict
from oracle.meshanything_oracle import Oracle
policy, path = sys.argv[2], sys.argv[3]
cfg = MAConfig.tiny(); sd = synthetic_state_dict(cfg)
d = dict(np.load(path))
o = Oracle(cfg, sd, policy)
x = torch.from_numpy(d["x"])
out = o.forward(x, suppress_eos=True)
res = {k: v.numpy() for k, v in out.items()}
res["tf_logits"] = torch.stack([o.teacher_forced_logits(out["prefix"][r:r + 1], out["tokens"][r]) for r in range(x.shape[0])]).numpy()
res["sampled"] = o.generate(out["prefix"][:1], sampling=True, uniforms=d["u"], suppress_eos=True).numpy()
np.savez(path + ".out.npz", **res)
"""


@pytest.mark.parametrize("policy", ["fp32", "bf16"])
def test_oracle_cuda_equals_oracle_cpu(policy, tmp_path):
    from oracle.meshanything_oracle import Oracle, normalize_pc, verify_greedy_stream, verify_sampled_stream
    cfg = MAConfig.tiny()
    sd = synthetic_state_dict(cfg)
    g = torch.Generator().manual_seed(21)
    d = torch.randn(2, cfg.n_points, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    cloud = torch.cat([d * (0.3 + 0.7 * torch.rand(2, cfg.n_points, 1, generator=g)), d], -1).numpy().astype(np.float32)
    x = np.stack([normalize_pc(c) for c in cloud])
    u = np.random.default_rng(4).random((1, cfg.max_new_tokens)).astype(np.float32)
    path = str(tmp_path / "io.npz")
    np.savez(path, x=x, u=u)
    try:
        r = subprocess.run([sys.executable, "-c", _CPU_SIDE, REPO, policy, path], capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired:
        pytest.fail("the CPU oracle (tiny shape, fresh interpreter, one thread) did not finish in 120 s: the host is unusable")
    assert r.returncode == 0, r.stderr[-2000:]
    a = {k: torch.from_numpy(v) for k, v in np.load(path + ".out.npz").items()}

    dev = Oracle(cfg, sd, policy, device="cuda")
    tol = 2e-5 if policy == "fp32" else 2e-2          # bf16: one rounding flip of an activation moves a logit by ~1e-2
    b = dev.forward(torch.from_numpy(x), suppress_eos=True)
    assert all(v.device.type == "cpu" for v in b.values()), "public oracle calls hand their results back on the CPU"
    assert float((a["point_feature"] - b["point_feature"]).abs().max()) < tol
    assert float((a["prefix"] - b["prefix"]).abs().max()) < 4 * tol
    ambiguous = 0
    for r_ in range(2):
        # the CPU oracle's greedy stream is a valid greedy decode under the torch-ROCm arithmetic, with the same logits
        v = verify_greedy_stream(dev, a["prefix"][r_:r_ + 1], a["tokens"][r_], 10 * tol, suppress_eos=True)
        assert v["hard"] == [], v
        ambiguous += v["ambiguous"]
        lb = dev.teacher_forced_logits(a["prefix"][r_:r_ + 1], a["tokens"][r_])
        assert float((a["tf_logits"][r_] - lb).abs().max()) < 10 * tol
    if ambiguous == 0:                                # no near-tie anywhere: the two devices must have decided identically
        assert torch.equal(a["tokens"], b["tokens"]) and torch.equal(a["ids"], b["ids"])
        assert int((torch.nan_to_num(a["coords"], nan=9.0) != torch.nan_to_num(b["coords"], nan=9.0)).sum()) <= 2
    # the CPU oracle's sampled stream (injected uniforms) is a valid draw under the torch-ROCm distributions
    v = verify_sampled_stream(dev, a["prefix"][:1], a["sampled"][0], u[0], tol=10 * tol, suppress_eos=True)
    assert v["hard"] == [], v
