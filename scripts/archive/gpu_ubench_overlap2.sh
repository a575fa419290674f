#!/bin/bash
# round 3: two-sequence launch-overlap microbenchmark (scripts/ubench_overlap2.hip) -> gpurun_out/r03_ubench_overlap2.log
mkdir -p gpurun_out; export TMPDIR=/tmp
{
hipcc --offload-arch=gfx950 -O3 scripts/ubench_overlap2.hip -o /tmp/ub2 2>&1 | grep -i "error" | head -5
timeout 240 /tmp/ub2 ${UB_STEPS:-40} 2>&1
echo "rc=$?"
} > gpurun_out/r03_ubench_overlap2.log 2>&1
tail -c 6000 gpurun_out/r03_ubench_overlap2.log
