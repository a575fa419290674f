#!/bin/bash
# Round-3 dense-phase work: correctness of the new attention kernel, then A/B of attention kernels and GEMM variants.
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== kernel tests: attention (both generations), gemm_tile"
timeout 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention or gemm_bf16_tile" 2>&1 | tail -8
echo "== dense phases, attention first generation (attn_impl=1)"
timeout 300 python scripts/prof_dense.py --batches 16,64 --iters 3 --options attn_impl=1 2>&1 | grep -v amdgpu.ids
echo "== dense phases, attention second generation (attn_impl=2)"
timeout 300 python scripts/prof_dense.py --batches 16,64 --iters 3 --options attn_impl=2 2>&1 | grep -v amdgpu.ids
echo "== gemm variants"
timeout 400 python scripts/ubench_gemm.py ${GEMM_VARIANTS:-0,4,5,6,7} 2>&1 | grep -v amdgpu.ids
for v in ${DENSE_GEMM_VARIANTS:-4 5}; do
echo "== dense phases, attn_impl=2, gemm_variant=$v"
timeout 300 python scripts/prof_dense.py --batches 64 --iters 3 --options attn_impl=2,gemm_variant=$v 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r03_dense_ab.log 2>&1
tail -c 7000 gpurun_out/r03_dense_ab.log
