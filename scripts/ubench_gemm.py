#!/usr/bin/env python3
"""gemm_tile throughput on the dense phases' problem shapes and on a large square problem, per kernel variant."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshanything_amd import _lib
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.engine import Engine

lib = _lib.load()
eng = Engine(MAConfig.tiny(dtype=DTYPE_BF16))
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
shapes = [(16448, 768, 768), (16384, 768, 768), (16448, 1024, 1024), (16384, 1024, 1024), (16448, 4096, 1024), (16448, 1024, 4096), (16384, 1024, 4096), (16448, 3072, 1024),
          (16448, 2304, 768), (16448, 768, 3072), (16384, 768, 3072), (67648, 768, 768), (67648, 3072, 768), (262144, 1536, 768), (8192, 8192, 8192)]
if os.environ.get("GEMM_SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["GEMM_SHAPES"].split(",")]
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,2").split(",")]
for v in variants:
    eng.set_option("gemm_variant", v)
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        run = lambda: _lib.check(lib.ma_op_gemm_bf16(p(A), K, p(W), p(b), None, 0, None, 0, p(Cb), N, M, N, K, 0, st), None)
        for _ in range(3): run()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        it = 10
        ev[0].record()
        for _ in range(it): run()
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / it
        tf = 2.0 * M * N * K / ms / 1e9
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        print(f"[variant {v}] M {M:6d} N {N:5d} K {K:5d}: {ms*1e3:9.1f} us  {tf:7.1f} TFLOP/s ({tf/25:5.1f} % of 2500)  tiles128 {tiles}", flush=True)
