#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  taskset -c 0-7 timeout 1200 python -m pytest ${TESTS:-tests/test_gpu_fidelity.py tests/test_gpu_rccl.py tests/test_gpu_kernels.py} -m gpu -q -x -s ${KEXPR:+-k "$KEXPR"} 2>&1 | tail -60
} > gpurun_out/new_tests.log 2>&1
tail -c 8000 gpurun_out/new_tests.log
