#!/usr/bin/env python3
"""Round 3 diagnosis: sampled draws of the 350M-shape batched path that fall outside the oracle's CDF interval -- how far, at which
rank, and does the rate depend on the batch size (kernel forms) or only on the number of draws (bf16 noise)?"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.engine import Engine
from oracle.meshanything_oracle import Oracle, EOS
from test_gpu_pipeline import mouse_variants

B, n = 64, 96
cfg = MAConfig.full(dtype=DTYPE_BF16, max_batch=B)
sd = synthetic_state_dict(cfg)
oracle = Oracle(cfg, sd, "bf16", device="cuda")
o32 = Oracle(cfg, sd, "fp32", device="cuda")
eng = Engine(cfg); eng.load_weights(sd.items())
x = mouse_variants(os.path.join(R, "tests", "golden"), B)
prefix = torch.cat([oracle.process_point_feature(oracle.encode_latents(x[i:i + 16])) for i in range(0, B, 16)])
u = torch.rand(B, n, generator=torch.Generator().manual_seed(64))

def analyse(tag, toks, rows, orc):
    tot = 0; bad = []
    dist_hist = []
    for bi, b in enumerate(rows):
        with orc.on_device():
            lg = orc.teacher_forced_logits(prefix[b:b + 1], toks[bi])[:n].float().clone()
        lg[:, EOS] = float("-inf")
        topv, topi = torch.topk(lg, 50, dim=-1)
        p = torch.softmax(topv, -1)
        tail = torch.flip(torch.cumsum(torch.flip(p, [-1]), -1), [-1])
        keep = ~(tail <= 0.05); keep[:, 0] = True
        nk = keep.sum(-1)
        mask = torch.arange(50, device=lg.device)[None] < nk[:, None]
        pr = torch.softmax(torch.where(mask, topv, torch.full_like(topv, float("-inf"))), -1)
        cdf = torch.cumsum(pr, -1)
        tk = toks[bi].to(lg.device)
        is_tok = topi == tk[:, None]
        rank = torch.where(is_tok.any(-1), is_tok.float().argmax(-1), torch.full_like(nk, 99))
        r = rank.clamp(max=49)
        hi = cdf.gather(1, r[:, None])[:, 0]; lo = hi - pr.gather(1, r[:, None])[:, 0]
        uu = u[b].to(lg.device)
        d = torch.maximum(lo - uu, uu - hi).clamp(min=0)
        d = torch.where(rank < nk, d, torch.full_like(d, 9.0))
        tot += n
        for j in (d > 0.06).nonzero().flatten().tolist():
            bad.append((b, j, int(rank[j]), int(nk[j]), round(float(uu[j]), 4), round(float(lo[j]), 4), round(float(hi[j]), 4), round(float(d[j]), 4), round(float(topv[j, 0] - topv[j, 49]), 3)))
        dist_hist += d[d < 9].tolist()
    dh = np.array(dist_hist)
    print(f"[{tag}] {tot} draws: outside by >6e-2: {len(bad)};  distance quantiles of in-set tokens: 50% {np.quantile(dh, .5):.4f} 90% {np.quantile(dh, .9):.4f} 99% {np.quantile(dh, .99):.4f} max {dh.max():.4f}")
    for t in bad[:12]:
        print("    row %d step %d: rank %d of %d kept, u %.4f interval [%.4f, %.4f] distance %.4f, top1-top50 logit spread %.3f" % t)

toks, _ = eng.generate(prefix.cuda(), sampling=True, uniforms=u, max_new_tokens=n, suppress_eos=True)
analyse("B=64 vs bf16 oracle", toks, list(range(B)), oracle)
analyse("B=64 vs fp32 oracle", toks, list(range(B)), o32)
for Bs in (16, 4, 1):
    rows = list(range(0, 16))
    out = []
    for i in range(0, 16, Bs):
        t, _ = eng.generate(prefix[i:i + Bs].cuda(), sampling=True, uniforms=u[i:i + Bs], max_new_tokens=n, suppress_eos=True)
        out.append(t)
    analyse(f"rows 0-15 run as batches of {Bs} vs bf16 oracle", torch.cat(out), rows, oracle)
