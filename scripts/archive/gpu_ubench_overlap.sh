#!/bin/bash
# round 3: launch-overlap microbenchmark (scripts/ubench_overlap.hip) -> gpurun_out/r03_ubench_overlap.log
mkdir -p gpurun_out; export TMPDIR=/tmp
{
hipcc --offload-arch=gfx950 -O3 scripts/ubench_overlap.hip -o /tmp/ubench_overlap 2>&1 | tail -5
timeout 240 /tmp/ubench_overlap ${UB_STEPS:-40} 2>&1
echo "rc=$?"
} > gpurun_out/r03_ubench_overlap.log 2>&1
tail -c 5000 gpurun_out/r03_ubench_overlap.log
