#!/bin/bash
# round 4, call S: hunt the once-seen failure of test_gemm_256_tile: the kernel test file 20 times, every failure message kept
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/r04s_repeat.txt
for i in $(seq 1 20); do
  python -m pytest tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids > /tmp/run_$i.txt
  echo "run $i: $(grep -E 'passed|failed' /tmp/run_$i.txt | tail -1)" >> gpurun_out/r04s_repeat.txt
  if grep -q "failed" /tmp/run_$i.txt; then grep -E "^E  |^FAILED|Error" /tmp/run_$i.txt | cut -c1-400 | head -30 >> gpurun_out/r04s_repeat.txt; cp /tmp/run_$i.txt gpurun_out/r04s_fail_$i.txt; fi
done
cat gpurun_out/r04s_repeat.txt | cut -c1-300
