#!/bin/bash
# round 3, check 2: new kernel tests (GEMM variants, LayerNorm outlier), XCD-local head mapping A/B (repeated: boxes drift), debug-build run
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== kernel tests"
timeout 400 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "gemm_bf16_tile or layernorm_prologue or attention_bf16" 2>&1 | grep -E "one-pass|passed|failed|Error|gemm_tile, variant . M 16448|gemm_tile, variant . M 4112" | tail -30
echo "== bitwise tests of the fused launches with the XCD-local mapping (default on)"
timeout 400 python -m pytest tests/test_gpu_persist.py -x -q -k "fused_qkv or fall_back" 2>&1 | tail -3
echo "== batch-1 decode step: qkv_xcd_local 0 / 1 / 0 / 1"
timeout 600 python scripts/prof_step.py --steps 32 --options "qkv_xcd_local=0;qkv_xcd_local=1;qkv_xcd_local=0;qkv_xcd_local=1" 2>&1 | grep -v amdgpu.ids
echo "== debug build (MA_DEBUG=1: -O1 -g, asserts alive): tiny pipeline + kernel tests"
MA_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model_api.py -x -q -k "tiny or facade or reference_call or composition" 2>&1 | tail -4
MA_DEBUG=1 python -c "from meshanything_amd import _lib; print(_lib.LIB_PATH); print(_lib.load().ma_version().decode())"
} > gpurun_out/r03_check2.log 2>&1
tail -c 6000 gpurun_out/r03_check2.log
