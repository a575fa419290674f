"""Does a CU-hogging kernel on one stream run beside work on another (MI355X, ROCm 7)?  E1 trivial torch kernel, E2 the engine's
dense encode, E3 decode with eager launches, E4 decode through the hipGraph -- each started while 250 CUs are held for 300 ms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.engine import Engine
cfg = MAConfig.full(dtype=DTYPE_BF16, max_batch=1)
eng = Engine(cfg); eng.load_weights(synthetic_state_dict(cfg).items())
x = torch.from_numpy(dict(np.load("tests/golden/dataset.npz"))["mouse_norm"])[None].cuda()
_, prefix = eng.encode(x)
eng.generate(prefix, max_new_tokens=8, suppress_eos=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.zeros(1 << 20, device="cuda")
def run(label, fn, blocks=250):
    torch.cuda.synchronize()
    base = eng.get_option("chain_fallbacks")
    t0 = time.time()
    eng.occupy_cus(blocks, 300_000, stream=s1)
    with torch.cuda.stream(s2):
        fn()
        s2.synchronize()
    t1 = time.time()
    torch.cuda.synchronize()
    print(f"  {label} (hog {blocks} blocks): returned after {1e3 * (t1 - t0):.1f} ms; fallbacks {eng.get_option('chain_fallbacks') - base}", flush=True)
    eng.set_option("chain_resident", 1)
for blocks in (250, 128):
    run("E1 trivial torch kernel", lambda: a.add_(1), blocks)
    run("E2 engine.encode", lambda: eng.encode(x), blocks)
    eng.set_option("use_graph", 0)
    run("E3 generate 8 tokens, eager", lambda: eng.generate(prefix, max_new_tokens=8, suppress_eos=True), blocks)
    eng.set_option("use_graph", 1)
    run("E4 generate 8 tokens, graph", lambda: eng.generate(prefix, max_new_tokens=8, suppress_eos=True), blocks)
# hog with small LDS: 1024 blocks x 64 threads that do not exclude anyone
torch.cuda.synchronize(); t0 = time.time()
eng.occupy_cus(250, 300_000, stream=s1, lds_bytes=1024)
with torch.cuda.stream(s2):
    a.add_(1); s2.synchronize()
print(f"  E5 trivial kernel beside a hog with 1 KiB of LDS per block: {1e3 * (time.time() - t0):.1f} ms")
torch.cuda.synchronize()
