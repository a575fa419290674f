#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== kernels"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider --tb=short -x 2>&1 | tail -15
  echo "== pipeline tiny"; timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model_api.py -m gpu -q --no-header -p no:cacheprovider --tb=short -k "not full" -x 2>&1 | tail -40
} > gpurun_out/quick_check.log 2>&1
tail -c 3000 gpurun_out/quick_check.log
for B in ${BATCHES:-4 8 32 64}; do
  timeout 600 python scripts/prof_step.py --batch $B --steps 4 --options "gemv_rpw=1" 2>&1 | grep "len" 
done | tee gpurun_out/prof_batch.log
