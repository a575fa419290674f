#!/bin/bash
# round 4, call L: rows_prologue / pick with their requests up front; A/B of the fc2 K split and of the LayerNorm fold at 12 / 16 rows
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== kernel + pipeline tests of the batched path"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm_dec or rows_prologue or attn" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s -k "batch or mfma or v2_scale or large or sampl" 2>&1 | grep -v amdgpu.ids > gpurun_out/r04l_batched_tests.txt; tail -4 gpurun_out/r04l_batched_tests.txt; grep -E "^E  |^FAILED" gpurun_out/r04l_batched_tests.txt | head
echo "== steps"
timeout 300 python scripts/prof_step.py --batch 8 --steps 8 --options "mfma_fc2_ksplit=4;mfma_fc2_ksplit=2;mfma_fc2_ksplit=1" 2>&1 | grep "len"
for B in 12 16; do timeout 300 python scripts/prof_step.py --batch $B --steps 6 --options "use_graph=1;mfma_fold_fc1_max=16;mfma_fold_fc1_max=16,mfma_fold_qkv_max=16;mfma_fold_fc1_max=16,mfma_fold_qkv_max=16,mfma_ln_waves=8" 2>&1 | grep "len" ; done
timeout 300 python scripts/prof_step.py --batch 64 --steps 4 --options "use_graph=1" 2>&1 | grep "len"
