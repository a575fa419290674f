#!/bin/bash
# final-form attention below 16 rows: more waves per (row, head) block instead of split-KV + merge launch
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity"
  taskset -c 0-7 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "decode_attention_rows" 2>&1 | tail -4
  for B in 8 12 4; do
    echo "== B=$B"
    timeout 600 python scripts/prof_step.py --batch $B --steps 8 --options "attn_final_min_batch=1000;attn_final_min_batch=4,attn_final_waves=16;attn_final_min_batch=4,attn_final_waves=8;attn_final_min_batch=4,attn_final_waves=4" 2>&1 | grep -v amdgpu.ids
  done
} > gpurun_out/batched_ab2.log 2>&1
tail -c 9000 gpurun_out/batched_ab2.log
