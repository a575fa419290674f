#!/bin/bash
# kernel trace of the batch decode step on the rows-looped launches -> gpurun_out/r03_rf_kernel_stats.csv
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o d --output-format csv -- python $R/scripts/prof_step.py --batch ${BATCH:-8} --steps 8 --options "${OPTS:-rows_fused=1}" > $R/gpurun_out/r03_rf_prof.log 2>&1
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r03_rf_kernel_stats.csv; done
cd $R; grep "B=" gpurun_out/r03_rf_prof.log; head -12 gpurun_out/r03_rf_kernel_stats.csv | cut -c1-180
