"""Round-3 diagnostics on the GPU box: (A) where do tiny-shape oracle calls spend their time (CPU vs torch-ROCm, per phase)?
(B) does a CU-hogging kernel on one stream really run beside the engine's stream (two non-blocking streams)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", os.environ.get("DIAG_THREADS", "1"))
import numpy as np, torch
torch.set_num_threads(int(os.environ.get("DIAG_THREADS", "1")))
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.checkpoint import synthetic_state_dict
from oracle.meshanything_oracle import Oracle, normalize_pc, verify_greedy_stream

def T(label, fn):
    torch.cuda.synchronize(); t0 = time.time(); r = fn(); torch.cuda.synchronize(); print(f"  {label}: {time.time() - t0:.3f} s", flush=True); return r

cfg = MAConfig.tiny(); sd = synthetic_state_dict(cfg)
g = torch.Generator().manual_seed(21)
d = torch.randn(2, cfg.n_points, 3, generator=g); d = d / d.norm(dim=-1, keepdim=True)
x = torch.from_numpy(np.stack([normalize_pc(c) for c in torch.cat([d * 0.8, d], -1).numpy().astype(np.float32)]))
for dev in ("cpu", "cuda", "cpu", "cuda"):
    print(f"[A] oracle on {dev} (threads {torch.get_num_threads()})")
    o = T("construct", lambda: Oracle(cfg, sd, "fp32", device=dev))
    lat = T("encode_latents", lambda: o.encode_latents(x))
    pre = T("process_point_feature", lambda: o.process_point_feature(lat))
    toks = T("generate 2 x 74", lambda: o.generate(pre, suppress_eos=True))
    T("teacher_forced_logits", lambda: o.teacher_forced_logits(pre[:1], toks[0]))
    T("verify_greedy_stream", lambda: verify_greedy_stream(o, pre[:1], toks[0], 1e-4, True))
    ids = o.postprocess_tokens(toks)
    T("detokenize", lambda: o.detokenize(ids, o.get_codes(ids), lat))
# raw op overhead
a = torch.randn(128, 128); b = torch.randn(128, 128)
t0 = time.time()
for _ in range(2000): c = a @ b
print(f"[A] 2000 x (128x128 @ 128x128) on the CPU: {time.time() - t0:.3f} s")
t0 = time.time()
for _ in range(2000): c = torch.nn.functional.layer_norm(a, (128,))
print(f"[A] 2000 x layer_norm(128x128) on the CPU: {time.time() - t0:.3f} s")
t0 = time.time()
for _ in range(2000): c = a + b
print(f"[A] 2000 x add on the CPU: {time.time() - t0:.3f} s")

print("[B] hog + engine on two non-blocking streams")
from meshanything_amd.engine import Engine
cfgf = MAConfig.full(dtype=DTYPE_BF16, max_batch=1)
eng = Engine(cfgf); eng.load_weights(synthetic_state_dict(cfgf).items())
mouse = torch.from_numpy(dict(np.load("tests/golden/dataset.npz"))["mouse_norm"])[None].cuda()
_, prefix = eng.encode(mouse)
want, _ = eng.generate(prefix, max_new_tokens=24, suppress_eos=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for variant in ("null+side", "side+side"):
    torch.cuda.synchronize()
    base = eng.get_option("chain_fallbacks")
    t0 = time.time()
    eng.occupy_cus(250, 300_000, stream=s1)
    if variant == "null+side":
        got, _ = eng.generate(prefix, max_new_tokens=24, suppress_eos=True)
    else:
        with torch.cuda.stream(s2):
            got, _ = eng.generate(prefix, max_new_tokens=24, suppress_eos=True)
    t1 = time.time()
    torch.cuda.synchronize()
    print(f"  {variant}: generate returned after {1e3 * (t1 - t0):.0f} ms (hog 300 ms), fallbacks {eng.get_option('chain_fallbacks') - base}, "
          f"chain_resident {eng.get_option('chain_resident')}, tokens equal {torch.equal(got.cpu(), want.cpu())}", flush=True)
    eng.set_option("chain_resident", 1)
