#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
{
  if [ -n "$PARITY" ]; then taskset -c 0-7 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_model_api.py -m gpu -q -x 2>&1 | tail -8; fi
  timeout 600 python scripts/prof_dense.py --batches ${BATCHES:-1,8,16,64} 2>&1 | grep -v amdgpu.ids
  echo "== rocprof kernel stats, B=${PB:-64} dense phases"
  cd /tmp; rm -rf /tmp/prof_d
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o d --output-format csv -- python $R/scripts/prof_dense.py --batches ${PB:-64} --iters 2 > $R/gpurun_out/prof_dense_rocprof.log 2>&1
  for f in $(find /tmp/prof_d -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/dense_kernel_stats.csv; done
  head -12 $R/gpurun_out/dense_kernel_stats.csv | cut -c1-200
} > gpurun_out/dense.log 2>&1
tail -c 6000 gpurun_out/dense.log
