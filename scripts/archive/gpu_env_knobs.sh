#!/bin/bash
# round 3: HIP runtime knobs that could move the per-launch cost of the decode chain (kernel arguments in device memory, hardware queues)
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for knob in "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=1" "HSA_XNACK=0" "HIP_FORCE_DEV_KERNARG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" ""; do
  echo "== env: ${knob:-default}"
  env $knob timeout 200 python scripts/prof_step.py --steps 32 --options "fuse_fc2=1" 2>&1 | grep -v "amdgpu.ids\|weights loaded"
done
} > gpurun_out/r03_env_knobs.log 2>&1
tail -c 5000 gpurun_out/r03_env_knobs.log
