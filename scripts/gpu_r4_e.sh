#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 ./scripts/ubench_gemm256 2>&1 | tee gpurun_out/ubench_gemm256_v2.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "gemm_256 or gemm_bf16_tile" 2>&1 | grep -E "^\[gemm256|passed|failed|^FAILED|^E  " | tee gpurun_out/r04e_gemm256.txt | tail -40
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04e_steps.txt
import sys, json, torch
sys.path.insert(0, ".")
import bench
from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F16
from meshanything_amd.checkpoint import synthetic_state_dict
from meshanything_amd.engine import Engine
sd = synthetic_state_dict(MAConfig.full(), init="diverse")
for name, dt in (("bf16", DTYPE_BF16), ("fp16", DTYPE_F16)):
    cfg = MAConfig.full(dtype=dt, max_batch=1)
    eng = Engine(cfg); eng.load_weights(sd.items())
    for L in (300, 3858, 7400):
        eng.profile_decode(L, 2)
        p = eng.profile_decode(L, 16)
        print(f"[{name}] kv {L}: step graph {p['step_ms_graph']*1e3:.1f} us eager {p['step_ms_eager']*1e3:.1f} us; weights {p['ms']['gemv']/p['launches']['gemv']*1e3:.2f} us/launch, cache {p['ms']['attn_decode']/p['launches']['attn_decode']*1e3:.2f} us/launch")
    eng.close()
cfg = MAConfig.full(dtype=DTYPE_BF16, max_batch=64)
eng = Engine(cfg); eng.load_weights(sd.items())
for mode in (1, 0, 1, 0):
    eng.set_option("gemm256", mode)
    print("gemm256 =", mode, json.dumps(bench.dense_phase_table(eng, cfg)))
eng.close()
PY
