#!/bin/bash
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for B in ${BATCHES:-8 64}; do
  rm -rf /tmp/pb
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python $R/scripts/prof_step.py --batch $B --steps 4 --options "gemv_rpw=1" > $R/gpurun_out/prof_b$B.log 2>&1
  for f in $(find /tmp/pb -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/batch${B}_kernel_stats.csv; done
  echo "== B=$B"; grep "len" $R/gpurun_out/prof_b$B.log | tail -4
  head -9 $R/gpurun_out/batch${B}_kernel_stats.csv | cut -c1-130
done
