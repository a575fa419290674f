#!/bin/bash
# round 4, call I: in-kernel timeline of the batched (8 rows) decode step, kernel stats of the batch-8 bench, experimental-build persist tests
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== step timeline, 8 rows"
timeout 300 python scripts/trace_step.py --batch 8 --lens 300,3858 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_trace_b8.log; cat gpurun_out/r04_trace_b8.log
echo "== kernel stats of the batch-8 bench"
cd /tmp; rm -rf /tmp/prof8
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof8 -o r4 --output-format csv -- python $R/bench.py --batch 8 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r04_prof_b8.json 2> $R/gpurun_out/r04_prof_b8.log
for f in $(find /tmp/prof8 -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r04_bench_batch8_kernel_stats.csv; done
head -8 $R/gpurun_out/r04_bench_batch8_kernel_stats.csv | cut -c1-200
cd $R
echo "== persist tests, MA_EXPERIMENTAL build"
MA_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_persist.py tests/test_gpu_rows_fused.py -q -m gpu 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04_suite_experimental.txt
tail -3 gpurun_out/r04_suite_experimental.txt; grep -E "^E  |^FAILED" gpurun_out/r04_suite_experimental.txt | head
