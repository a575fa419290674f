#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof1
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o r1 --output-format csv -- python $R/scripts/prof_decode.py --tokens 400 --full-forward > $R/gpurun_out/prof1.log 2>&1
find /tmp/prof1 -name "*stats*" | head; 
for f in $(find /tmp/prof1 -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r1_kernel_stats.csv; done
head -40 $R/gpurun_out/r1_kernel_stats.csv
tail -5 $R/gpurun_out/prof1.log
