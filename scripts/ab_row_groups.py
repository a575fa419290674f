#!/usr/bin/env python3
"""Row groups of the batched decode step (option decode_groups): graph-replayed step time at a fixed cache length, per batch size
and group count, plus full generations.  350M shape, bf16, synthetic weights."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.checkpoint import synthetic_items
from meshanything_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="8,12,16,24,32,64")
ap.add_argument("--groups", default="1,2,3,4")
ap.add_argument("--lens", default="1000,3858")
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--generate", default="8,16")          # batch sizes of the full 800-face generations
ap.add_argument("--options", default="")
a = ap.parse_args()
Bs = [int(x) for x in a.batches.split(",") if x]
cfg = MAConfig.full(dtype=DTYPE_BF16, max_batch=max(Bs + [int(x) for x in a.generate.split(",") if x]))
eng = Engine(cfg)
eng.load_weights(synthetic_items(cfg))
for kv in a.options.split(","):
    if kv:
        k, v = kv.split("="); eng.set_option(k, int(v))
for B in Bs:
    eng.set_option("profile_batch", B)
    for L in [int(x) for x in a.lens.split(",")]:
        row = []
        for G in [int(x) for x in a.groups.split(",")]:
            eng.set_option("decode_groups", G)
            if eng.get_option("decode_groups") != G:
                continue
            best = min(eng.profile_decode(L, a.steps)["step_ms_graph"] for _ in range(3))
            row.append((G, best))
        base = row[0][1]
        print(f"B={B:3d} kv={L:5d}: " + "  ".join(f"G={G}: {ms * 1e3:7.1f} us ({B / ms:7.0f} tok/s, x{base / ms:4.2f})" for G, ms in row), flush=True)
g = torch.Generator().manual_seed(0)
for B in [int(x) for x in a.generate.split(",") if x]:
    prefix = torch.randn(B, cfg.cond_length, cfg.hidden, generator=g).cuda() * 0.5
    for G in (1, 2, 4):
        eng.set_option("profile_batch", B)
        eng.set_option("decode_groups", G)
        eng.generate(prefix, max_new_tokens=64, suppress_eos=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        toks, _ = eng.generate(prefix, suppress_eos=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"generate B={B} decode_groups={G} (effective {eng.get_option('decode_groups')}): {toks.shape[1]} tokens/row in {dt:.3f} s = {B * toks.shape[1] / dt:8.1f} face-tokens/s", flush=True)
