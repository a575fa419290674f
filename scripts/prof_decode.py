#!/usr/bin/env python3
"""Short generation for rocprofv3: `rocprofv3 --kernel-trace --stats -d <dir> -- python scripts/prof_decode.py --tokens 600`."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F32
from meshanything_amd.checkpoint import synthetic_items
from meshanything_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=600)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--full-forward", action="store_true")
ap.add_argument("--graph", type=int, default=1)
a = ap.parse_args()
cfg = MAConfig.full(dtype=DTYPE_BF16 if a.dtype == "bf16" else DTYPE_F32, use_graph=a.graph)
eng = Engine(cfg)
eng.load_weights(synthetic_items(cfg))
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dataset.npz"))
x = torch.from_numpy(d["mouse_norm"])[None].cuda()
torch.cuda.synchronize()
t0 = time.time()
if a.full_forward:
    out = eng.forward(x, suppress_eos=True, max_new_tokens=a.tokens)
else:
    lat, prefix = eng.encode(x)
    toks, _ = eng.generate(prefix, suppress_eos=True, max_new_tokens=a.tokens)
torch.cuda.synchronize()
print(f"{a.tokens} tokens in {time.time()-t0:.3f}s")
