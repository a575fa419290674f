#!/bin/bash
# round 4, call K: 8-wave LayerNorm-folded GEMM (one row per wave) and the final-form attention with its first round requested before the
# row length is known: unit tests, batched pipeline tests, timeline, steps with the wave-count A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== kernel + pipeline tests of the batched path"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm_dec or rows_prologue or attn" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s -k "batch or mfma or v2_scale or large" 2>&1 | grep -v amdgpu.ids > gpurun_out/r04k_batched_tests.txt; tail -4 gpurun_out/r04k_batched_tests.txt; grep -E "^E  |^FAILED" gpurun_out/r04k_batched_tests.txt | head
echo "== step timeline, 8 rows"
timeout 300 python scripts/trace_step.py --batch 8 --lens 300,3858 2>&1 | grep -v amdgpu.ids > gpurun_out/r04k_trace_b8.log; cat gpurun_out/r04k_trace_b8.log
echo "== steps"
for B in 4 8; do timeout 300 python scripts/prof_step.py --batch $B --steps 8 --options "mfma_ln_waves=4;mfma_ln_waves=8" 2>&1 | grep "len" ; done
for B in 12 16 64; do timeout 300 python scripts/prof_step.py --batch $B --steps 4 --options "use_graph=1" 2>&1 | grep "len" ; done
