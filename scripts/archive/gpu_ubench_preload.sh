#!/bin/bash
# round 3: does kernarg preload (-mllvm -amdgpu-kernarg-preload-count=16: the first 16 argument dwords arrive in SGPRs with the wave instead of
# through s_load) shorten a dependent launch?  The launch-chain microbenchmark built both ways.
mkdir -p gpurun_out; export TMPDIR=/tmp
{
hipcc --offload-arch=gfx950 -O3 scripts/ubench_overlap.hip -o /tmp/ub_plain 2>&1 | grep -i error
hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 scripts/ubench_overlap.hip -o /tmp/ub_preload 2>&1 | grep -i error
for rep in 1 2; do
echo "== plain build"; timeout 120 /tmp/ub_plain 40 2>&1 | grep -A1 "mode 0\|mode 4" | grep -v "^--" | head -8
echo "== kernarg preload build"; timeout 120 /tmp/ub_preload 40 2>&1 | grep -A1 "mode 0\|mode 4" | grep -v "^--" | head -8
done
} > gpurun_out/r03_ubench_preload.log 2>&1
tail -c 5000 gpurun_out/r03_ubench_preload.log
