#!/bin/bash
# batch-1 chain: bitwise parity of the fused launches + step timing at 4 cache lengths + a 2048-token generation
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  taskset -c 0-7 timeout 900 python -m pytest tests/test_gpu_persist.py -m gpu -q -x -s -k "${TESTS:-fused or bitwise or fc2}" 2>&1 | grep -v "^$" | tail -18
  timeout 600 python scripts/prof_step.py --steps 16 --options "${OPTS:-use_graph=1}" --gen 2048 2>&1 | grep -v amdgpu.ids
} > gpurun_out/b1_check.log 2>&1
tail -c 6000 gpurun_out/b1_check.log
