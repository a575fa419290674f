#!/usr/bin/env python3
"""Soak of the 8-row two-launch layer: whole generations at 1600 faces (cache to 14 659 positions), counting what the engine's health counters saw --
sweeps that gave up, generations that fell back, scalar sweeps a vector look had to finish, the error word of the last fall-back.
  python scripts/stress_rows_b8.py [GENERATIONS] [FACES]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.engine import Engine
from conftest import mouse_variants, GOLDEN

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
faces = int(sys.argv[2]) if len(sys.argv) > 2 else 1600
# argv[3] = "forced": the form in which round 5's one fall-back was seen (tests/test_gpu_long_context.py::test_config5_batched_deep_cache): teacher-forced
# along the reference anchor's stream, logits of the steps from 7800 on kept (logits_first_step) -- the step graph then carries the logits copy
# argv[4] = rows_attn_early (6 default: q/k/v sweep by scalar loads; 3: by vector loads -- the control)
forced_form = len(sys.argv) > 3 and sys.argv[3] == "forced"
early = int(sys.argv[4]) if len(sys.argv) > 4 else 6
cfg = MAConfig.full(dtype=DTYPE_BF16, max_batch=8, n_max_faces=faces)
eng = Engine(cfg)
eng.load_weights(synthetic_state_dict(cfg, init="diverse").items())
_, prefix = eng.encode(mouse_variants(GOLDEN, 8).cuda())
eng.set_option("rows_attn_early", early)
kw = {}
if forced_form:
    import numpy as np
    anchor = np.load(os.path.join(GOLDEN, "full_anchor_long.npz"))
    forced = torch.from_numpy(anchor["long_tokens"][:cfg.max_new_tokens].astype("int64"))
    assert forced.shape[0] == cfg.max_new_tokens, "the long anchor covers 1600 faces only"
    kw = dict(forced_tokens=forced[None].expand(8, -1).contiguous(), return_logits=True, logits_first_step=7800)
print(f"form: {'forced tokens + logits from step 7800' if forced_form else 'free-running'}; rows_attn_early = {early}; fuse_rows_attn {eng.get_option('fuse_rows_attn')} fuse_rows_mlp {eng.get_option('fuse_rows_mlp')}", flush=True)
ref = None
for i in range(n):
    torch.cuda.synchronize(); t0 = time.time()
    out = eng.generate(prefix, suppress_eos=True, **kw)
    toks = out[0]
    del out
    torch.cuda.synchronize(); dt = time.time() - t0
    if ref is None:
        ref = toks.clone()
    print(f"generation {i}: {toks.shape[1]} steps x 8 rows in {dt:.2f} s ({8 * toks.shape[1] / dt:.0f} tok/s); same tokens as generation 0: {torch.equal(ref, toks)}; "
          f"timeouts {eng.get_option('xchg_timeouts')} fallbacks {eng.get_option('chain_fallbacks')} last code {eng.get_option('xchg_last_code')} "
          f"scalar sweeps rescued {eng.get_option('scalar_sweep_rescues')} slow blocks {eng.get_option('slow_blocks')} (max {eng.get_option('slow_block_max_us')} us) "
          f"resident {eng.get_option('chain_resident')}; first give-up ever: code {eng.get_option('xchg_first_giveup_code')} block/wave {eng.get_option('xchg_first_giveup_block'):#x} "
          f"polls {eng.get_option('xchg_first_giveup_polls')}", flush=True)
