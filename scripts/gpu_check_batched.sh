#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  taskset -c 0-7 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -m gpu -q -x -s -k "decode_attention_rows or large_batches or batched" 2>&1 | grep -v "^$" | tail -25
  timeout 900 python bench.py --batch 8 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/b8.json 2> gpurun_out/b8.err; cut -c1-300 gpurun_out/b8.json; tail -2 gpurun_out/b8.err
} > gpurun_out/check_batched.log 2>&1
tail -c 5000 gpurun_out/check_batched.log
