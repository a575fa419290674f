#!/usr/bin/env python3
"""Diagnostic for the fused 8-row launches (rows_attn.hpp / rows_mlp.hpp): which configuration disagrees with which, where, and is it stable?
Runs the same 8-row greedy generation several times per configuration and prints, for every pair of runs, the first (step, rows) whose
logits differ bitwise."""
import os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.engine import Engine
from conftest import mouse_variants, GOLDEN

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
from meshanything_amd.config import DTYPE_F16
cfg = MAConfig.full(dtype=DTYPE_F16 if (len(sys.argv) > 2 and sys.argv[2] == "fp16") else DTYPE_BF16, max_batch=8)
eng = Engine(cfg)
eng.load_weights(synthetic_state_dict(cfg, init="diverse").items())
_, prefix = eng.encode(mouse_variants(GOLDEN, 8).cuda())
runs = {}


def run(tag, attn, mlp, graph=1, reps=2, ln2=0):
    eng.set_option("fuse_rows_attn", attn); eng.set_option("fuse_rows_mlp", mlp); eng.set_option("use_graph", graph); eng.set_option("rows_mlp_ln2", ln2)
    for r in range(reps):
        t, _, g = eng.generate(prefix, max_new_tokens=n, suppress_eos=True, return_logits=True)
        runs[f"{tag}#{r}"] = (t.cpu(), g.clone())
    print(f"{tag}: timeouts {eng.get_option('xchg_timeouts')} fallbacks {eng.get_option('chain_fallbacks')} resident {eng.get_option('chain_resident')}", flush=True)


run("unfused", 0, 0, reps=2)
run("attn", 1, 0, reps=3)
run("mlp", 0, 1, reps=3)
run("both", 1, 1, reps=2)
run("both-ln2", 1, 1, reps=2, ln2=1)
run("mlp-ln2", 0, 1, reps=1, ln2=1)
eng.set_option("use_graph", 1)
names = list(runs)
ref = runs["unfused#0"][1]
for a in names:
    g = runs[a][1]
    d = (g.view(torch.int32) != ref.view(torch.int32))
    if not d.any():
        print(f"{a:16s} == unfused#0 on all {n} steps")
        continue
    step = int(d.any(dim=2).any(dim=0).nonzero()[0])
    rows = d[:, step].any(dim=1).nonzero().flatten().tolist()
    dd = (g[:, step] - ref[:, step]).abs()
    per_row = [int(d[b].any(dim=1).nonzero()[0]) if d[b].any() else -1 for b in range(8)]
    print(f"{a:16s} first differing step per row {per_row}")
    print(f"{a:16s} != unfused#0 from step {step}, rows {rows}: max abs {float(dd.max()):.3e}, {int(d[:, step].sum())} of {d[:, step].numel()} logits differ at that step; "
          f"tokens equal up to step {int((runs[a][0] != runs['unfused#0'][0]).any(dim=0).nonzero()[0]) if (runs[a][0] != runs['unfused#0'][0]).any() else n}")
# repeatability inside each configuration
for tag in ("unfused", "attn", "mlp", "both", "both-ln2"):
    same = all(torch.equal(runs[f"{tag}#0"][1].view(torch.int32), runs[k][1].view(torch.int32)) for k in names if k.startswith(tag + "#"))
    print(f"{tag}: repeats bit-identical: {same}")
