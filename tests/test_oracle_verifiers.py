"""The stream verifiers of oracle/ are themselves test infrastructure the GPU suite leans on: the vectorised per-draw decision
(`classify_sampled_draws`, used for 64-row batches at the 350M shape) must agree with the scalar walk of `verify_sampled_stream`
(one Python sort of the vocabulary per step) on every case -- exact draws, draws moved to a neighbouring interval, draws far away,
peaked rows where the top-p cut falls inside the top-k."""
import numpy as np
import torch

from oracle.meshanything_oracle import Oracle, classify_sampled_draws


def _scalar_decision(lg, tok, u, tol):
    V = lg.shape[0]
    kept, probs = Oracle.topk_topp_filter(lg)
    exact = Oracle.sample_from(kept, probs, u) == tok
    ok = exact
    kl = kept.tolist()
    if not ok and tok in kl:
        c = np.concatenate([[0.0], np.cumsum(probs.double().numpy())])
        i = kl.index(tok)
        ok = (c[i] - tol) <= u <= (c[i + 1] + tol)
    if not ok:
        order = sorted(range(V), key=lambda q: (-float(lg[q]), q))[:min(50, V)]
        pk = torch.softmax(lg[order].double(), dim=0).numpy()
        tail = np.cumsum(pk[::-1])[::-1]
        nkept = len(kl)
        for nk in (nkept - 1, nkept + 1):
            if nk < 1 or nk > len(order) or tok not in order[:nk]:
                continue
            r = nkept if nk > nkept else nkept - 1
            if abs(float(tail[r]) - 0.05) > tol:
                continue
            alt = order[:nk]
            c = np.concatenate([[0.0], np.cumsum(torch.softmax(lg[alt].double(), dim=0).numpy())])
            i = alt.index(tok)
            if (c[i] - tol) <= u <= (c[i + 1] + tol):
                ok = True
                break
    return exact, ok


def test_vectorised_sampling_verifier_equals_the_scalar_walk():
    g = torch.Generator().manual_seed(0)
    V, N = 300, 240
    logits = torch.randn(N, V, generator=g) * 2.0
    logits[::3] *= 3.0                                              # peaked rows: top-p removes part of the top-k
    logits[5, 7] = float("-inf")                                    # a suppressed token
    u = torch.rand(N, generator=g)
    toks = []
    for j in range(N):
        kept, probs = Oracle.topk_topp_filter(logits[j])
        du = [0.0, 0.01, -0.01, 0.2, 0.05][j % 5]
        toks.append(Oracle.sample_from(kept, probs, float(min(max(float(u[j]) + du, 0.0), 0.999999))))
    toks[11] = int(torch.argmin(logits[11]))                        # a token outside the top-k
    toks = torch.tensor(toks)
    for tol in (1e-4, 2e-2, 6e-2):
        c = classify_sampled_draws(logits, toks, u, tol)
        ref = [_scalar_decision(logits[j], int(toks[j]), float(u[j]), tol) for j in range(N)]
        assert c["exact"].tolist() == [bool(e) for e, _ in ref]
        assert c["ok"].tolist() == [bool(o) for _, o in ref], tol
        assert not c["ok"][11]
    assert 0.3 * N < int(c["exact"].sum()) < N                      # the cases really are mixed
