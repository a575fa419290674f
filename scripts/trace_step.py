#!/usr/bin/env python3
"""In-kernel timeline of one decode step: where does a launch's time go?  (ma_trace_decode, 100 MHz ticks = 10 ns)"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F32
from meshanything_amd.checkpoint import synthetic_items
from meshanything_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--lens", default="300,3800,7400")
ap.add_argument("--options", default="")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--dist", action="store_true", help="also print the distribution over blocks of every stamp of the fused launches")
a = ap.parse_args()
cfg = MAConfig.full(dtype=DTYPE_BF16 if a.dtype == "bf16" else DTYPE_F32, max_batch=a.batch)
eng = Engine(cfg)
eng.load_weights(synthetic_items(cfg))
for kv in a.options.split(","):
    if kv:
        k, v = kv.split("="); eng.set_option(k, int(v))
eng.set_option("profile_batch", a.batch)
names = ["embed", "qkv", "attn", "oproj", "fc1", "fc2", "lmhead"]
for L in [int(x) for x in a.lens.split(",")]:
    t = eng.trace_decode(L)
    ticks, kinds, blocks = t["ticks"].astype(np.int64), t["kinds"], t["blocks"]
    rows = {}
    prev_end = None
    step0 = None
    for i in range(len(kinds)):
        tk = ticks[i, :blocks[i]]
        t0, t1, t2, t3 = tk[:, 0], tk[:, 1], tk[:, 2], tk[:, 3]
        ok3 = t3 > 0
        start, end = t0.min(), (t3[ok3].max() if ok3.any() else t0.max())
        if step0 is None: step0 = start
        gap = (start - prev_end) if prev_end is not None else 0
        prev_end = end
        r = rows.setdefault(kinds[i], [])
        x_ready = np.median((t1 - t0)[t1 > 0]) if (t1 > 0).any() else 0
        w_ready = np.median((t2 - t0)[t2 > 0]) if (t2 > 0).any() else 0
        r.append([gap, t0.max() - start, x_ready, w_ready, np.median((t3 - t0)[ok3]) if ok3.any() else 0, end - start])
    total = (prev_end - step0) / 100.0
    print(f"== len {L}: step {total:.1f} us (first block start -> last block end, eager, traced)")
    print(f"{'kind':8s} {'n':>3s} {'gap_before':>10s} {'ramp':>7s} {'x_staged':>9s} {'w_done':>7s} {'blk_life':>9s} {'kernel':>7s}   (us, mean over launches; x/w/life = median over blocks)")
    for k in sorted(rows):
        m = np.mean(np.array(rows[k], dtype=np.float64), axis=0) / 100.0
        print(f"{names[k]:8s} {len(rows[k]):3d} {m[0]:10.2f} {m[1]:7.2f} {m[2]:9.2f} {m[3]:7.2f} {m[4]:9.2f} {m[5]:7.2f}")
    if a.dist:
        for i in range(len(kinds)):
            if blocks[i] != 256 or i not in (len(kinds) // 2, len(kinds) // 2 + 1):
                continue                                   # one first-half and one second-half launch from the middle of the step
            tk = ticks[i, :256]
            for j, nm in ((1, "stamp1"), (2, "stamp2"), (3, "stamp3")):
                d = (tk[:, j] - tk[:, 0])[tk[:, j] > 0] / 100.0
                if len(d):
                    q = np.percentile(d, [0, 10, 50, 90, 100])
                    print(f"   launch {i} ({names[kinds[i]]}) {nm}: {len(d)} blocks, us from the block's start: min {q[0]:.2f} p10 {q[1]:.2f} p50 {q[2]:.2f} p90 {q[3]:.2f} max {q[4]:.2f}")
