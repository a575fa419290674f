"""Is the tiny bf16 sampling mismatch noise or a bug of the new dense attention?  Same generation with attn_impl = 1 and 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.engine import Engine
from oracle.meshanything_oracle import Oracle, verify_sampled_stream, verify_greedy_stream
from test_gpu_pipeline import clouds
cfg = MAConfig.tiny(dtype=DTYPE_BF16, max_batch=4)
sd = synthetic_state_dict(cfg)
ora = Oracle(cfg, sd, "bf16", device="cuda")
eng = Engine(cfg); eng.load_weights(sd.items())
x = clouds(cfg, [11, 12])
prefix = ora.process_point_feature(ora.encode_latents(x))
u = np.random.default_rng(5).random((2, cfg.max_new_tokens)).astype(np.float32)
res = {}
for impl in (1, 2):
    eng.set_option("attn_impl", impl)
    toks, _ = eng.generate(prefix.cuda(), sampling=True, uniforms=torch.from_numpy(u), suppress_eos=True)
    res[impl] = toks.cpu()
    one, _ = eng.generate(prefix.cuda(), max_new_tokens=1, suppress_eos=True)
    lg = torch.stack([eng.read_logits(r).clone() for r in range(2)]).cpu()
    res[(impl, "lg")] = lg
    for b in range(2):
        v = verify_sampled_stream(ora, prefix[b:b + 1], toks[b].cpu(), u[b], tol=2e-2, suppress_eos=True)
        print(f"attn_impl {impl} row {b}: exact {v['exact']} ambiguous {v['ambiguous']} hard {v['hard']}")
    # prefill logits vs the oracle's
    ol = torch.stack([ora.lm_head(ora.opt_layers(ora.embed_prefix(prefix[b:b + 1]), None)[0, -1]) for b in range(2)])
    print(f"attn_impl {impl}: prefill logits vs bf16-policy oracle: max abs {float((lg - ol).abs().max()):.4f}")
print("tokens equal between impls:", torch.equal(res[1], res[2]), " first-step logits max diff:", float((res[(1, 'lg')] - res[(2, 'lg')]).abs().max()))
lat1 = None
for impl in (1, 2):
    eng.set_option("attn_impl", impl)
    lat, pre = eng.encode(x.cuda())
    if lat1 is None: lat1 = (lat.clone(), pre.clone())
    else: print("encoder latents / prefix, impl 1 vs 2: max abs", float((lat - lat1[0]).abs().max()), float((pre - lat1[1]).abs().max()))
    ol = ora.encode_latents(x)
    print(f"attn_impl {impl}: latents vs bf16-policy oracle max abs {float((lat.cpu() - ol).abs().max()):.4f}")
