"""Soak of the fused decode launches in the setting where their lone give-ups were seen (rounds 5 and 6: only in full-suite runs -- one process that
creates and destroys many engines): N rounds of { new engine (fp32 / bf16 / fp16 in turn), weights, encode, one teacher-forced generation with
returned logits (the failing test's call), one free-running generation, destroy }.  Per round: the health counters, whether the two streams are what the
first round of that policy produced.  GPU box; output kept as profiles/r06_soak_fused_engine_churn.txt."""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F16, DTYPE_F32
from meshanything_amd.engine import Engine
from meshanything_amd.checkpoint import synthetic_state_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
KEYS = ("chain_fallbacks", "xchg_timeouts", "xchg_descheduled", "scalar_sweep_rescues", "slow_blocks", "slow_block_max_us", "chain_resident")
x = torch.from_numpy(np.load(os.path.join(REPO, "tests", "golden", "dataset.npz"))["mouse_norm"])[None].cuda()
POL = (("fp32", DTYPE_F32), ("bf16", DTYPE_BF16), ("fp32", DTYPE_F32), ("fp16", DTYPE_F16))
sds, first, bad = {}, {}, 0
t_all = time.time()
for i in range(N):
    name, dt = POL[i % len(POL)]
    faces = 1600 if i % 6 == 5 else 800
    cfg = MAConfig.full(dtype=dt, n_max_faces=faces, max_batch=1)
    key = (name, faces)
    if key not in sds:
        sds[key] = synthetic_state_dict(cfg, init="diverse")
    t0 = time.time()
    eng = Engine(cfg)
    eng.load_weights(sds[key].items())
    _, prefix = eng.encode(x)
    free, _ = eng.generate(prefix, suppress_eos=True)
    forced = free.clone()
    toks, lengths, logits = eng.generate(prefix, suppress_eos=True, forced_tokens=forced, return_logits=True)
    ok_forced = bool(torch.equal(toks, forced))
    del logits
    free2, _ = eng.generate(prefix, suppress_eos=True)
    same = bool(torch.equal(free2, free))
    if key in first:
        same = same and bool(torch.equal(free.cpu(), first[key]))
    else:
        first[key] = free.cpu()
    h = {k: eng.get_option(k) for k in KEYS}
    clean = h["chain_fallbacks"] == 0 and h["xchg_timeouts"] == 0 and h["chain_resident"] == 1 and same and ok_forced
    bad += 0 if clean else 1
    print(f"round {i:3d} {name} {faces:4d} faces: 3 generations of {free.shape[1]} tokens in {time.time() - t0:5.1f} s; streams as the first round's {same}, forced walk {ok_forced}; "
          + ", ".join(f"{k} {v}" for k, v in h.items()) + ("" if clean else "   <-- NOT CLEAN"), flush=True)
    eng.close()
    del eng, prefix, free, free2, toks, forced
print(f"{N} rounds, {3 * N} generations in {time.time() - t_all:.0f} s: {bad} round(s) not clean")
