#!/usr/bin/env python3
"""A/B of an engine option of the 8-row launches (default: rows_attn.hpp's request placements, option rows_attn_early): bitwise equality of every step's
logits with the five-launch layer, then the graph-replayed step at three cache depths, two passes.
  python scripts/ab_rows_early.py VALUES [STEPS] [OPTION]        e.g.  3,5 700    or    0,11,12,18 300 rows_mlp_prefetch"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F16
from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.engine import Engine
from conftest import mouse_variants, GOLDEN

modes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3,5").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 700
OPT = sys.argv[3] if len(sys.argv) > 3 else "rows_attn_early"
for dt, name in ((DTYPE_BF16, "bf16"), (DTYPE_F16, "fp16")):
    cfg = MAConfig.full(dtype=dt, max_batch=8)
    eng = Engine(cfg)
    eng.load_weights(synthetic_state_dict(cfg, init="diverse").items())
    _, prefix = eng.encode(mouse_variants(GOLDEN, 8).cuda())
    eng.set_option("fuse_rows_attn", 0); eng.set_option("fuse_rows_mlp", 0)
    t0, _, g0 = eng.generate(prefix, max_new_tokens=n, suppress_eos=True, return_logits=True)
    eng.set_option("fuse_rows_attn", 1); eng.set_option("fuse_rows_mlp", 1)
    for m in modes:
        eng.set_option(OPT, m)
        t1, _, g1 = eng.generate(prefix, max_new_tokens=n, suppress_eos=True, return_logits=True)
        same = torch.equal(g0.view(torch.int32), g1.view(torch.int32)) and torch.equal(t0, t1)
        print(f"[{name}] {OPT}={m}: {n} steps x 8 rows bitwise equal to the five-launch layer: {same}; timeouts {eng.get_option('xchg_timeouts')} fallbacks {eng.get_option('chain_fallbacks')} "
              f"resident {eng.get_option('chain_resident')}", flush=True)
        del g1
    del g0
    if name == "bf16":
        eng.set_option("profile_batch", 8)
        for rep in range(2):
            for kv in (600, 3858, 7300):
                row = []
                for m in modes:
                    eng.set_option(OPT, m)
                    p = eng.profile_decode(kv, 3)
                    row.append(f"{m}: {1e3 * p['step_ms_graph']:7.1f}")
                print(f"kv {kv:5d} | " + " | ".join(row), flush=True)
    del eng
    torch.cuda.empty_cache()
