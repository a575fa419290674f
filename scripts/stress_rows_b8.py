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
cfg = MAConfig.full(dtype=DTYPE_BF16, max_batch=8, n_max_faces=faces)
eng = Engine(cfg)
eng.load_weights(synthetic_state_dict(cfg, init="diverse").items())
_, prefix = eng.encode(mouse_variants(GOLDEN, 8).cuda())
ref = None
for i in range(n):
    torch.cuda.synchronize(); t0 = time.time()
    toks, _ = eng.generate(prefix, suppress_eos=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    if ref is None:
        ref = toks.clone()
    print(f"generation {i}: {toks.shape[1]} steps x 8 rows in {dt:.2f} s ({8 * toks.shape[1] / dt:.0f} tok/s); same tokens as generation 0: {torch.equal(ref, toks)}; "
          f"timeouts {eng.get_option('xchg_timeouts')} fallbacks {eng.get_option('chain_fallbacks')} last code {eng.get_option('xchg_last_code')} "
          f"scalar sweeps rescued {eng.get_option('scalar_sweep_rescues')} slow blocks {eng.get_option('slow_blocks')} (max {eng.get_option('slow_block_max_us')} us) "
          f"resident {eng.get_option('chain_resident')}", flush=True)
