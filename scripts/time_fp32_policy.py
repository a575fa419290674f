"""One warm mesh under the fp32 ("exact") policy with the fused batch-1 launches on and off (GPU box): the two token streams must be the same; the
decode step at three cache lengths both ways.  Output kept under profiles/ (r06_ab_fp32_policy_fused_chain.txt)."""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from meshanything_amd.config import MAConfig, DTYPE_F32
from meshanything_amd.engine import Engine
from meshanything_amd.checkpoint import synthetic_state_dict

cfg = MAConfig.full(dtype=DTYPE_F32, n_max_faces=800, max_batch=1)
eng = Engine(cfg)
eng.load_weights(synthetic_state_dict(cfg, init="diverse").items())
x = torch.from_numpy(np.load(os.path.join(REPO, "tests", "golden", "dataset.npz"))["mouse_norm"])[None].cuda()
print("effective options:", {k: eng.get_option(k) for k in ("chain_resident", "resident_blocks", "fuse_qkv_attn", "fuse_oproj_fc1", "fuse_fc2")})
toks = {}
for fuse in (1, 0, 1, 0):
    eng.set_option("fuse_qkv_attn", fuse); eng.set_option("fuse_oproj_fc1", fuse)
    eng.forward(x, suppress_eos=True, max_new_tokens=64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    o = eng.forward(x, suppress_eos=True)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    n = o["tokens"].shape[1]
    toks[fuse] = o["tokens"].cpu()
    print(f"fused launches {fuse}: {n} tokens in {t:.3f} s = {n / t:.1f} face-tokens/s; fall-backs {eng.get_option('chain_fallbacks')}, expiries {eng.get_option('xchg_timeouts')}")
print("token streams identical:", bool(torch.equal(toks[0], toks[1])))
for L in (300, 3858, cfg.max_seq - 80):
    row = {}
    for fuse in (0, 1):
        eng.set_option("fuse_qkv_attn", fuse); eng.set_option("fuse_oproj_fc1", fuse)
        eng.profile_decode(L, 2)
        row[fuse] = eng.profile_decode(L, 16)["step_ms_graph"] * 1e3
    print(f"decode step at kv {L}: five launches {row[0]:.1f} us | two fused launches {row[1]:.1f} us | ratio {row[1] / row[0]:.3f}")
