#!/bin/bash
# quick loop: kernel parity + tiny pipeline parity + decode-step timing + rocprof of a short generation
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
{
  echo "== kernels"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider --tb=short -x 2>&1 | tail -25
  echo "== pipeline tiny"; timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model_api.py -m gpu -q --no-header -p no:cacheprovider --tb=short -k "not full" -x 2>&1 | tail -40
} > gpurun_out/quick_check.log 2>&1
tail -c 3000 gpurun_out/quick_check.log
echo "== step timing"
timeout 600 python scripts/prof_step.py ${PROF_ARGS:-} > gpurun_out/prof_step.log 2>&1
grep -v "amdgpu.ids" gpurun_out/prof_step.log | tail -20
echo "== rocprof"
cd /tmp; rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o q --output-format csv -- python $R/scripts/prof_step.py --options "gemv_rpw=1" --steps 2 --gen 1024 > $R/gpurun_out/prof_quick.log 2>&1
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/quick_kernel_stats.csv; done
head -12 $R/gpurun_out/quick_kernel_stats.csv | cut -c1-200
tail -2 $R/gpurun_out/prof_quick.log
