#!/usr/bin/env python3
"""BASELINE.md section 4's CPU baseline, in full: the fp32 oracle (a CPU port of the reference's arithmetic: plain PyTorch,
KV cache grown by torch.cat per step as transformers 4.39.3 does) on the host cores of the box, for BASELINE.json configs[1] -- pc_examples/
mouse.npy, 350M shape, greedy, 800-face cap, eos suppressed: encode / prefill / EVERY one of the 7 201 decode steps / detokenize, with the
step time at contexts 300, 3 800 and 7 400 (mean of the 32 steps around each) and the core count.  One JSON object on stdout and in
gpurun_out/cpu_baseline_full.json; bench.py keeps its bounded sample and cites the committed copy under profiles/.
    python scripts/cpu_baseline_full.py [threads=min(16, cores)] [max_steps=7201]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(16, os.cpu_count() or 1)
for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[k] = str(threads)

import numpy as np      # noqa: E402
import torch            # noqa: E402

from meshanything_amd.checkpoint import synthetic_state_dict     # noqa: E402
from meshanything_amd.config import MAConfig, DTYPE_F32           # noqa: E402
from oracle.meshanything_oracle import Oracle                     # noqa: E402


def main():
    torch.set_num_threads(threads)
    cfg = MAConfig.full(dtype=DTYPE_F32)
    max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else cfg.max_new_tokens - 1
    sd = synthetic_state_dict(cfg, init="diverse")
    x = torch.from_numpy(np.load(os.path.join(REPO, "tests", "golden", "dataset.npz"))["mouse_norm"])[None]
    o = Oracle(cfg, sd, "fp32")
    load1 = os.getloadavg()
    t0 = time.time()
    lat = o.encode_latents(x)
    prefix = o.process_point_feature(lat)
    t_enc = time.time() - t0
    t0 = time.time()
    cache = [None] * cfg.layers
    h = o.opt_layers(o.embed_prefix(prefix), cache)
    lg = o.lm_head(h[0, -1])
    lg[1] = float("-inf")
    toks = [int(torch.argmax(lg))]
    t_prefill = time.time() - t0
    stamps = [time.time()]
    for n in range(1, max_steps + 1):
        e = o.embed_tokens(torch.tensor([toks[-1]]), torch.tensor([n]))
        h = o.opt_layers(e[None], cache)
        lg = o.lm_head(h[0, -1])
        lg[1] = float("-inf")
        toks.append(int(torch.argmax(lg)))
        stamps.append(time.time())
        if n % 500 == 0:
            print(f"[cpu baseline] step {n}/{max_steps}: {stamps[-1] - stamps[0]:.0f} s", file=sys.stderr, flush=True)
    t_dec = stamps[-1] - stamps[0]
    dt = np.diff(np.array(stamps))
    at = {}
    for ctx in (300, 3800, 7400):
        n = ctx - cfg.cond_length                       # the step whose cache holds `ctx` positions
        if 16 <= n and n + 16 <= len(dt):
            at[str(ctx)] = round(float(dt[n - 16:n + 16].mean()) * 1e3, 2)
    t0 = time.time()
    full = len(toks) == cfg.max_new_tokens
    t_detok = None
    if full:
        ids = o.postprocess_tokens(torch.tensor([toks]))
        o.detokenize(ids, o.get_codes(ids), lat)
        t_detok = time.time() - t0
    total = t_enc + t_prefill + t_dec + (t_detok or 0.0)
    res = {"what": "fp32 oracle (CPU port of the reference arithmetic, torch CPU) on BASELINE.json configs[1]: mouse.npy, 350M, greedy, 800-face cap, eos suppressed",
           "threads": threads, "host_cores": os.cpu_count(), "loadavg_before": [round(v, 1) for v in load1], "loadavg_after": [round(v, 1) for v in os.getloadavg()],
           "tokens": len(toks), "tokens_distinct": len(set(toks)), "complete": full,
           "encode_s": round(t_enc, 2), "prefill_s": round(t_prefill, 2), "decode_s": round(t_dec, 1), "detokenize_s": None if t_detok is None else round(t_detok, 2),
           "step_ms_at_context": at, "sec_per_mesh": round(total, 1), "face_tokens_per_s": round(len(toks) / total, 2),
           "step_ms_first_100": round(float(dt[:100].mean()) * 1e3, 2), "step_ms_last_100": round(float(dt[-100:].mean()) * 1e3, 2)}
    line = json.dumps(res)
    print(line)
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "cpu_baseline_full.json"), "w") as f:
        f.write(line + "\n")


if __name__ == "__main__":
    main()
