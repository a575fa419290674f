#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for B in ${BS:-8 12 16 6}; do
    timeout 600 python scripts/prof_step.py --batch $B --steps 8 --options "${OPTS:-mfma_fold_ln=0;mfma_fold_ln=1,mfma_fold_fc1_max=16,mfma_fold_qkv_max=0;mfma_fold_ln=1,mfma_fold_fc1_max=16,mfma_fold_qkv_max=16}" 2>&1 | grep -v amdgpu.ids | grep "len" | sed 's/  per-class.*//'
  done
} > gpurun_out/fold_ln2.log 2>&1
tail -c 6000 gpurun_out/fold_ln2.log
