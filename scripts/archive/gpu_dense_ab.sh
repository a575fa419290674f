#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  taskset -c 0-7 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm_bf16_tile or test_gemm" 2>&1 | tail -4
  for o in ${OPTS:-gemm_xcd_swizzle=0 gemm_xcd_swizzle=1}; do
    timeout 600 python scripts/prof_dense.py --batches ${BATCHES:-16,64} --options "$o" 2>&1 | grep -v amdgpu.ids
  done
} > gpurun_out/dense_ab.log 2>&1
tail -c 5000 gpurun_out/dense_ab.log
