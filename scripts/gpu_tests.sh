#!/bin/bash
# parity only: kernels + tiny pipeline + facade (fast), optionally the full-size pipeline (FULL=1)
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== kernels"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -15
  echo "== pipeline tiny + facade"; timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model_api.py -m gpu -q --no-header -p no:cacheprovider --tb=short -k "not full" 2>&1 | tail -40
  if [ -n "$FULL" ]; then echo "== pipeline full"; timeout 2400 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --no-header -p no:cacheprovider --tb=short -s -k "full" 2>&1 | tail -40; fi
} > gpurun_out/tests.log 2>&1
tail -c 4000 gpurun_out/tests.log
