#!/usr/bin/env python3
"""Decode-step timing at several cache lengths, with A/B engine options.  `python scripts/prof_step.py [--gen N]`."""
import argparse, os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F32
from meshanything_amd.checkpoint import synthetic_items
from meshanything_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--gen", type=int, default=0, help="also generate this many tokens (for rocprofv3 kernel traces)")
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--options", default="gemv_rpw=1;gemv_rpw=2")
ap.add_argument("--faces", type=int, default=800)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--lens", default="", help="cache lengths of the profile_decode sweeps (default: 300, 1000, 3800 and the maximum)")
ap.add_argument("--no-profile", action="store_true", help="skip the profile_decode sweeps (PMC passes: only the generated steps, cache 257 .. 257 + gen)")
a = ap.parse_args()
cfg = MAConfig.full(dtype=DTYPE_BF16 if a.dtype == "bf16" else DTYPE_F32, n_max_faces=a.faces, max_batch=a.batch)
eng = Engine(cfg)
t0 = time.time()
eng.load_weights(synthetic_items(cfg))
print(f"weights loaded in {time.time()-t0:.1f}s", flush=True)
eng.set_option("profile_batch", a.batch)
lens = [int(v) for v in a.lens.split(",")] if a.lens else [300, 1000, 3800, cfg.max_seq - 3 * a.steps - 16]
for opt in a.options.split(";"):
    for kv in opt.split(","):
        if kv:
            k, v = kv.split("=")
            eng.set_option(k, int(v))
    for L in ([] if a.no_profile else lens):
        eng.profile_decode(L, 2)                      # warm (graph capture, clocks)
        p = eng.profile_decode(L, a.steps)
        per = {k: round(v / a.steps * 1e3, 1) for k, v in p["ms"].items() if p["launches"][k]}
        print(f"[B={a.batch} {opt}] len {L:5d}: step graph {p['step_ms_graph']*1e3:7.1f} us  eager {p['step_ms_eager']*1e3:7.1f} us  per-class(us, event-bracketed eager) {per}", flush=True)
if a.gen:
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dataset.npz"))
    x = torch.from_numpy(d["mouse_norm"])[None].expand(a.batch, -1, -1).contiguous().cuda()
    lat, prefix = eng.encode(x)
    torch.cuda.synchronize(); t0 = time.time()
    toks, _ = eng.generate(prefix, suppress_eos=True, max_new_tokens=a.gen)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"generate {a.gen} tokens: {dt:.3f}s = {a.gen/dt:.1f} tok/s", flush=True)
