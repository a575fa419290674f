#!/usr/bin/env python3
"""Stress of the two-stream prefill (csrc/engine.hip prefill: the rows behind the 256-row tiles as a chain of their own on a second stream).
Its claim is that NOTHING of the main chain depends on when the tail chain runs, and nothing of the tail chain on anything but the per-layer event: so the
logits must be the one-stream form's bit for bit whatever the timing.  This script perturbs the timing and compares every run:
  * B = 64, 16 and 40 samples, bf16 and fp16; per case a set of P different prefixes (so that whatever a run leaves in the workspace is NOT what the next
    run should compute), each first run in the one-stream form (prefill_tail = 0): the reference logits of the prefill's token and 3 decode steps;
  * then N two-stream runs cycling through the prefixes while a second stream parks 16 .. 128 workgroups that hold a whole CU each for 50 .. 2000 us at a
    time (the tail chain's kernels find room at other moments) and a third stream copies 256 MB blocks through HBM;
  * every run's logits compared bitwise with the reference of its prefix.
Usage: python scripts/stress_prefill_tail.py [runs_per_case=120]    (output kept as profiles/r06_stress_prefill_tail.txt)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F16
from meshanything_amd.engine import Engine
from meshanything_amd.checkpoint import synthetic_state_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
P, STEPS = 4, 4
side, copy_s = torch.cuda.Stream(), torch.cuda.Stream()
src = torch.empty(1 << 28, dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty_like(src)
rng = torch.Generator().manual_seed(7)
total = bad = 0
t_all = time.time()
for name, dt in (("bf16", DTYPE_BF16), ("fp16", DTYPE_F16)):
    for B in (64, 16, 40):
        cfg = MAConfig.full(dtype=dt, max_batch=B)
        eng = Engine(cfg)
        eng.load_weights(synthetic_state_dict(cfg, init="diverse").items())
        g = torch.Generator().manual_seed(100 + B)
        prefixes = [(torch.randn(B, cfg.num_latents + 1, cfg.hidden, generator=g) * 0.5).cuda() for _ in range(P)]
        eng.set_option("prefill_tail", 0)
        refs = []
        for px in prefixes:
            t, _, lg = eng.generate(px, max_new_tokens=STEPS, suppress_eos=True, return_logits=True)
            refs.append((t.clone(), lg.clone()))
        eng.set_option("prefill_tail", 2)
        case_bad = 0
        t0 = time.time()
        release = torch.zeros(1, dtype=torch.int32).pin_memory()
        for i in range(N):
            k = int(torch.randint(0, P, (1,), generator=rng))
            mode = i % 4
            if mode in (1, 3):
                cus = int(torch.randint(16, 129, (1,), generator=rng)); us = int(torch.randint(50, 2001, (1,), generator=rng))
                eng.occupy_cus(cus, us, stream=side, release=release)
            if mode in (2, 3):
                with torch.cuda.stream(copy_s):
                    dst.copy_(src, non_blocking=True)
            t, _, lg = eng.generate(prefixes[k], max_new_tokens=STEPS, suppress_eos=True, return_logits=True)
            ok = bool(torch.equal(t, refs[k][0])) and bool(torch.equal(lg.view(torch.int32), refs[k][1].view(torch.int32)))
            if not ok:
                case_bad += 1
                d = (lg - refs[k][1]).abs()
                print(f"  MISMATCH {name} B={B} run {i} prefix {k} mode {mode}: max abs logit difference {float(d.max()):.4e} in row {int(d.amax(dim=(1, 2)).argmax())}", flush=True)
        torch.cuda.synchronize()
        total += N; bad += case_bad
        print(f"[{name} B={B:2d}] {N} two-stream prefills (+ {STEPS - 1} decode steps) under CU hogs / HBM copies on two other streams, {P} prefixes in rotation: "
              f"{case_bad} differ from the one-stream form  ({time.time() - t0:.1f} s; fall-backs {eng.get_option('chain_fallbacks')})", flush=True)
        eng.close()
print(f"total: {total} runs, {bad} mismatches, {time.time() - t_all:.0f} s")
