#!/bin/bash
# persistent decode step: parity against the launch chain + A/B timing + edge timeline
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  taskset -c 0-7 timeout 900 python -m pytest tests/test_gpu_persist.py -m gpu -q -x -s 2>&1 | tail -40
} > gpurun_out/persist.log 2>&1
tail -c 6000 gpurun_out/persist.log
