#!/bin/bash
# One gpurun call: kernel parity, full-size pipeline parity, smoke, bench, rocprof kernel trace of the bench command.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
{
  echo "== device"; python -c "import torch;print(torch.cuda.get_device_name(0), torch.cuda.device_count())"; nproc
  echo "== kernels"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -30
  echo "== pipeline"; timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model_api.py -m gpu -q --no-header -p no:cacheprovider --tb=short -s 2>&1 | tail -80
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5
} > gpurun_out/check.log 2>&1
tail -c 5000 gpurun_out/check.log
echo "== bench"
timeout 1200 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 3000 gpurun_out/bench.json; tail -c 1500 gpurun_out/bench.err
echo "== rocprof"
cd /tmp; rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r1 --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched-table > $R/gpurun_out/prof.log 2>&1
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r1_bench_kernel_stats.csv; done
head -30 $R/gpurun_out/r1_bench_kernel_stats.csv
