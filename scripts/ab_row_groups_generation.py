"""Whole generations of 64 rows (top-k / top-p sampling, 800 faces: BASELINE configs[2]) with option decode_groups = 1 | 0 | 2: seconds and tokens/s, identical-stream
check between repeats of one setting.  GPU box."""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.engine import Engine
from meshanything_amd.checkpoint import synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = MAConfig.full(dtype=DTYPE_BF16, n_max_faces=800, max_batch=B)
eng = Engine(cfg)
eng.load_weights(synthetic_state_dict(cfg, init="diverse").items())
g = torch.Generator().manual_seed(3)
prefix = (torch.randn(B, cfg.num_latents + 1, cfg.hidden, generator=g) * 0.5).cuda()
eng.generate(prefix, sampling=True, seed=5, suppress_eos=True, max_new_tokens=128)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 7202
for opt, ce in ((1, 64), (2, 64), (2, 8), (2, 3), (1, 8), (2, 16)):
    eng.set_option("decode_groups", opt)
    eng.generate(prefix, sampling=True, seed=5, suppress_eos=True, max_new_tokens=64)      # warm: graphs of this layout
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks, _ = eng.generate(prefix, sampling=True, seed=5, suppress_eos=True, max_new_tokens=N, check_every=ce)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"decode_groups={opt} check_every={ce}: {B} x {toks.shape[1]} tokens in {dt:.2f} s = {B * toks.shape[1] / dt:.0f} tok/s; fall-backs {eng.get_option('chain_fallbacks')}", flush=True)
