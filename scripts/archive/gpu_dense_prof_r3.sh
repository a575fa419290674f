#!/bin/bash
# kernel trace of the dense phases at batch 64 (attn_impl / gemm_variant from $OPTS) -> gpurun_out/r03_dense_kernel_stats.csv
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o d --output-format csv -- python $R/scripts/prof_dense.py --batches ${BATCHES:-64} --iters 2 --options ${OPTS:-attn_impl=2,gemm_variant=6} > $R/gpurun_out/r03_dense_prof.log 2>&1
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r03_dense_kernel_stats.csv; done
cd $R; grep dense gpurun_out/r03_dense_prof.log; head -16 gpurun_out/r03_dense_kernel_stats.csv | cut -c1-200
