#!/bin/bash
# round 4: test_gemm_256_tile alone, 40 times, first on a fresh box (the once-seen failure happened in the first GPU work of a box)
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/r04_gemm256_repeat.txt
for i in $(seq 1 40); do
  python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm_256_tile or gemm_dec" 2>&1 | grep -v amdgpu.ids > /tmp/g_$i.txt
  echo "run $i: $(grep -E 'passed|failed' /tmp/g_$i.txt | tail -1)" >> gpurun_out/r04_gemm256_repeat.txt
  if grep -q "failed" /tmp/g_$i.txt; then grep -E "^E  |^FAILED" /tmp/g_$i.txt | cut -c1-500 | head -20 >> gpurun_out/r04_gemm256_repeat.txt; fi
done
sort gpurun_out/r04_gemm256_repeat.txt | uniq -c | sort -rn | head -5 | cut -c1-300
grep -c passed gpurun_out/r04_gemm256_repeat.txt
