#!/bin/bash
# A/B of engine options on the decode step (graph replay, 4 cache lengths) + a short parity run.  OPTS="a=1;b=2,c=3" bash scripts/gpu_ab.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity (kernels + tiny pipeline)"
  taskset -c 0-7 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -m gpu -q -x -k "not full and not v2_scale" 2>&1 | tail -6
  echo "== A/B"
  timeout 900 python scripts/prof_step.py --steps ${STEPS:-16} --options "${OPTS:-pf_dist=0;pf_dist=1}" ${GEN:+--gen $GEN} 2>&1 | grep -v amdgpu.ids
} > gpurun_out/ab.log 2>&1
tail -c 7000 gpurun_out/ab.log
