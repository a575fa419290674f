"""The 8-row two-launch layer under engine churn (the setting of round 5's lone fall-back): N rounds of { new engine (bf16 / fp16 in turn, 8 rows, 800 or 1 600 faces),
weights, two whole generations (free-running + teacher-forced along it), destroy }.  Per round the health counters.  GPU box; output appended to
profiles/r06_soak_fused_engine_churn.txt."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F16
from meshanything_amd.engine import Engine
from meshanything_amd.checkpoint import synthetic_state_dict
N = int(sys.argv[1]) if len(sys.argv) > 1 else 15
KEYS = ("chain_fallbacks", "xchg_timeouts", "xchg_descheduled", "scalar_sweep_rescues", "slow_blocks", "slow_block_max_us", "chain_resident")
sds, first, bad = {}, {}, 0
t_all = time.time()
for i in range(N):
    name, dt = (("bf16", DTYPE_BF16), ("fp16", DTYPE_F16))[i % 2]
    faces = 1600 if i % 3 == 2 else 800
    cfg = MAConfig.full(dtype=dt, n_max_faces=faces, max_batch=8)
    key = (name, faces)
    if key not in sds:
        sds[key] = synthetic_state_dict(cfg, init="diverse")
    t0 = time.time()
    eng = Engine(cfg)
    eng.load_weights(sds[key].items())
    g = torch.Generator().manual_seed(11)
    prefix = (torch.randn(8, cfg.num_latents + 1, cfg.hidden, generator=g) * 0.5).cuda()
    free, _ = eng.generate(prefix, suppress_eos=True)
    toks, _ = eng.generate(prefix, suppress_eos=True, forced_tokens=free)
    same = bool(torch.equal(toks, free))
    if key in first:
        same = same and bool(torch.equal(free.cpu(), first[key]))
    else:
        first[key] = free.cpu()
    h = {k: eng.get_option(k) for k in KEYS}
    clean = h["chain_fallbacks"] == 0 and h["xchg_timeouts"] == 0 and h["chain_resident"] == 1 and same
    bad += 0 if clean else 1
    print(f"round {i:3d} {name} 8 rows {faces:4d} faces: 2 generations of {free.shape[1]} tokens in {time.time() - t0:5.1f} s; streams as expected {same}; fused rows_attn {eng.get_option('fuse_rows_attn')}; "
          + ", ".join(f"{k} {v}" for k, v in h.items()) + ("" if clean else "   <-- NOT CLEAN"), flush=True)
    eng.close()
    del eng, prefix, free, toks
print(f"{N} rounds, {2 * N} generations of 8 rows in {time.time() - t_all:.0f} s: {bad} round(s) not clean")
