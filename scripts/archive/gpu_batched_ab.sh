#!/bin/bash
# batched decode path: final-form attention (>= 16 rows) vs split + merge launch; row-per-wave vs block-per-chunk attention at 8 rows;
# MFMA vs row-parallel GEMV crossover at 4 rows.  Parity first.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity"
  taskset -c 0-7 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -m gpu -q -x -k "decode_attention_rows or large_batches or batched_mfma" 2>&1 | tail -8
  echo "== B=64 final vs split"
  timeout 600 python scripts/prof_step.py --batch 64 --steps 8 --options "attn_final_min_batch=1000;attn_final_min_batch=16" 2>&1 | grep -v amdgpu.ids
  echo "== B=16 final vs split"
  timeout 600 python scripts/prof_step.py --batch 16 --steps 8 --options "attn_final_min_batch=1000;attn_final_min_batch=16" 2>&1 | grep -v amdgpu.ids
  echo "== B=8 rowwave vs block"
  timeout 600 python scripts/prof_step.py --batch 8 --steps 8 --options "attn_rowwave=1;attn_rowwave=0" 2>&1 | grep -v amdgpu.ids
  echo "== B=4 mfma vs gemv rows"
  timeout 600 python scripts/prof_step.py --batch 4 --steps 8 --options "mfma_min_batch=4;mfma_min_batch=65" 2>&1 | grep -v amdgpu.ids
} > gpurun_out/batched_ab.log 2>&1
tail -c 9000 gpurun_out/batched_ab.log
