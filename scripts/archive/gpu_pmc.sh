#!/bin/bash
# HBM traffic of the decode kernels from the PMC counters: separate passes per counter, kernel-trace only (MI355X guide)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 150 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p --output-format csv -- python $R/scripts/prof_step.py --options "gemv_rpw=4" --steps 2 --gen 96 > $R/gpurun_out/pmc_$C.log 2>&1
  tail -2 $R/gpurun_out/pmc_$C.log
done
python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_summary.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
ls /tmp/pmc_FETCH_SIZE/* | head
