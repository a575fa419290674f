#!/usr/bin/env python3
"""Calibration, not product: what the vendor's GEMM library (hipBLASLt / rocBLAS behind torch.nn.functional.linear) reaches on the dense phases'
problem shapes on THIS box, next to the package's own tile (csrc/gemm256.hpp through ma_op_gemm_bf16: bias epilogue, 16-bit output).  Says how much
of the distance to the 2.5 PFLOP/s bf16 peak is this silicon's at K = 768 ... 4096 and how much is the kernel's."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshanything_amd import _lib
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.engine import Engine

lib = _lib.load()
eng = Engine(MAConfig.tiny(dtype=DTYPE_BF16))
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
# (M, N, K): prefill at 64 samples (q/k/v, out_proj, fc1, fc2), the point encoder's self-attention layers, the detokenizer, a large square problem
shapes = [(16448, 3072, 1024), (16448, 1024, 1024), (16448, 4096, 1024), (16448, 1024, 4096), (16448, 2304, 768), (16448, 768, 768), (16448, 3072, 768), (16448, 768, 3072),
          (67648, 768, 768), (67648, 3072, 768), (8192, 8192, 8192)]
it = 20


def timed(run):
    for _ in range(3):
        run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(it):
        run()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / it


print(f"{'M':>6s} {'N':>5s} {'K':>5s} | {'gemm256 us':>10s} {'TFLOP/s':>8s} | {'library us':>10s} {'TFLOP/s':>8s} | {'library, no bias':>16s} | ratio own/library")
for (M, N, K) in shapes:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    bb = b.to(torch.bfloat16)
    Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    own = timed(lambda: _lib.check(lib.ma_op_gemm_bf16(p(A), K, p(W), p(b), None, 0, None, 0, p(Cb), N, M, N, K, 0, st), None))
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    libt = timed(lambda: torch.addmm(bb, A, W.t(), out=out))
    libn = timed(lambda: torch.mm(A, W.t(), out=out))
    err = float((out.float() - (Cb.float() - b[None, :])).abs().max())
    fl = 2.0 * M * N * K / 1e9
    print(f"{M:6d} {N:5d} {K:5d} | {own * 1e3:10.1f} {fl / own:8.1f} | {libt * 1e3:10.1f} {fl / libt:8.1f} | {libn * 1e3:8.1f} {fl / libn:7.1f} | {libt / own:5.2f}   (max |own - library| {err:.3f})", flush=True)
