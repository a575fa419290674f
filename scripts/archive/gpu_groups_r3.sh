#!/bin/bash
# round 3: row groups of the batched decode step -- parity (groups == their own batches, bitwise) and A/B of step time / generations
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== parity: tests/test_gpu_pipeline.py -k 'row_groups or large_batches or batched'"
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -k "row_groups or large_batches or batched" 2>&1 | tail -6
echo "== step time per batch size and group count (graph replay, best of 3 x ${STEPS:-16} steps)"
timeout 900 python scripts/ab_row_groups.py --steps ${STEPS:-16} ${AB_ARGS} 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r03_row_groups.log 2>&1
tail -c 5000 gpurun_out/r03_row_groups.log
