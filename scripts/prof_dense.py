#!/usr/bin/env python3
"""Dense-phase throughput (encoder + prefix, prefill, detokenizer) at a batch of shapes: ms and TFLOP/s against the bf16 MFMA peak.
Algorithmic FLOPs per shape from SURVEY.md 8d: encoder 108.5 + prefix projections 0.8 + to_shape_latents 61 (inside process_point_feature),
prefill 158.5, detokenizer 115.6 GFLOP."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.checkpoint import synthetic_items
from meshanything_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="1,8,16,64")
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--options", default="", help="engine options, e.g. gemm_xcd_swizzle=0")
ap.add_argument("--phases", default="encode,prefill,detok", help="which phases run inside the loop (a kernel trace of ONE phase: --phases detok)")
a = ap.parse_args()
batches = [int(b) for b in a.batches.split(",")]
phases = set(a.phases.split(","))
cfg = MAConfig.full(dtype=DTYPE_BF16, max_batch=max(batches))
eng = Engine(cfg)
eng.load_weights(synthetic_items(cfg))
for kv in a.options.split(","):
    if kv:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
        print(f"option {k} = {v}", flush=True)
PEAK = 2500.0
GF = {"encode+prefix": 108.5 + 0.8, "prefill": 158.5, "detokenize": 115.6}
g = torch.Generator().manual_seed(0)
for B in batches:
    d = torch.randn(B, cfg.n_points, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    x = torch.cat([d * 0.9, d], dim=-1).half().cuda()
    ids = torch.randint(0, cfg.codebook_size, (B, cfg.n_max_faces * 9), generator=g).cuda()
    res = {}
    if "encode" not in phases:
        lat, prefix = eng.encode(x)
    for it in range(a.iters + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        if "encode" in phases:
            lat, prefix = eng.encode(x)
        ev[1].record()
        if "prefill" in phases:
            toks, _ = eng.generate(prefix, max_new_tokens=1, suppress_eos=True)       # prefill + the first pick
        ev[2].record()
        if "detok" in phases:
            coords = eng.detokenize(ids, lat)
        ev[3].record()
        torch.cuda.synchronize()
        if it:
            for k, i, ph in (("encode+prefix", 0, "encode"), ("prefill", 1, "prefill"), ("detokenize", 2, "detok")):
                if ph in phases:
                    res.setdefault(k, []).append(ev[i].elapsed_time(ev[i + 1]))
    line = f"[dense B={B:3d}]"
    tot_ms, tot_gf = 0.0, 0.0
    for k, v in res.items():
        ms = float(np.median(v)); tf = GF[k] * B / ms
        tot_ms += ms; tot_gf += GF[k] * B
        line += f" {k} {ms:8.2f} ms = {tf:6.1f} TFLOP/s ({tf / PEAK * 100:4.1f} %) |"
    line += f" all {tot_ms:8.2f} ms = {tot_gf / tot_ms:6.1f} TFLOP/s ({tot_gf / tot_ms / PEAK * 100:4.1f} % of {PEAK:.0f})"
    print(line, flush=True)
