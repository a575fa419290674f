#!/bin/bash
# One gpurun call: full-size parity, bench, rocprof kernel trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== kernels"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider --tb=line 2>&1 | tail -15
  echo "== pipeline full"; timeout 2400 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --no-header -p no:cacheprovider --tb=short -s -k "full" 2>&1 | tail -80
} > gpurun_out/check_b.log 2>&1
tail -c 5000 gpurun_out/check_b.log
echo "== bench"
timeout 1200 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
tail -c 3000 gpurun_out/bench_b.json; tail -c 1500 gpurun_out/bench_b.err
