#!/bin/bash
# round 4, call J: batched decode kernels with every operand requested up front: unit tests, batched pipeline tests, timeline, batch-8 steps
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== kernel + pipeline tests of the batched path"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm_dec or rows_prologue" 2>&1 | grep -v amdgpu.ids | tail -5
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s -k "batch or mfma or v2_scale or large" 2>&1 | grep -v amdgpu.ids > gpurun_out/r04j_batched_tests.txt; tail -4 gpurun_out/r04j_batched_tests.txt; grep -E "^E  |^FAILED" gpurun_out/r04j_batched_tests.txt | head
echo "== step timeline, 8 rows"
timeout 300 python scripts/trace_step.py --batch 8 --lens 300,3858 2>&1 | grep -v amdgpu.ids > gpurun_out/r04j_trace_b8.log; cat gpurun_out/r04j_trace_b8.log
echo "== steps"
for B in 4 8 16; do timeout 300 python scripts/prof_step.py --batch $B --steps 8 --options "use_graph=1" 2>&1 | grep "len" ; done
