#!/usr/bin/env python3
"""A/B of an engine option of the batch-1 decode launches: bitwise equality of every step's logits across the values (batch 1 and 2, bf16 and fp16),
then the graph-replayed step at three cache depths, three passes.   python scripts/ab_batch1_option.py OPTION VALUES [STEPS]   e.g. qkv_fast_rounds 0,1 1500"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F16
from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.engine import Engine
from conftest import mouse_variants, GOLDEN

OPT = sys.argv[1]
vals = [int(v) for v in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
for dt, name in ((DTYPE_BF16, "bf16"), (DTYPE_F16, "fp16")):
    cfg = MAConfig.full(dtype=dt, max_batch=2)
    eng = Engine(cfg)
    eng.load_weights(synthetic_state_dict(cfg, init="diverse").items())
    for B in (1, 2):
        _, prefix = eng.encode(mouse_variants(GOLDEN, B).cuda())
        ref = None
        for v in vals:
            eng.set_option(OPT, v)
            t, _, g = eng.generate(prefix, max_new_tokens=n, suppress_eos=True, return_logits=True)
            if ref is None:
                ref = (t, g)
                continue
            same = torch.equal(ref[1].view(torch.int32), g.view(torch.int32)) and torch.equal(ref[0], t)
            print(f"[{name}] batch {B}: {n} steps, {OPT}={v} bitwise equal to {OPT}={vals[0]}: {same}; timeouts {eng.get_option('xchg_timeouts')} fallbacks "
                  f"{eng.get_option('chain_fallbacks')}", flush=True)
            del g
        del ref
    if name == "bf16":
        for rep in range(3):
            for kv in (300, 3800, 7400):
                row = []
                for v in vals:
                    eng.set_option(OPT, v)
                    p = eng.profile_decode(kv, 3)
                    row.append(f"{v}: {1e3 * p['step_ms_graph']:7.1f}")
                print(f"kv {kv:5d} | " + " | ".join(row), flush=True)
    del eng
    torch.cuda.empty_cache()
