#!/bin/bash
# round 3, check 3: the BASELINE-config shape tests (configs[2]: 64 rows sampling; configs[4]: 1600 faces, batch 8) and the wall time of `python bench.py`
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== tests: v2_scale (configs[2] and configs[4] at their own shapes)"
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -s -k "v2_scale" 2>&1 | grep -v "^$" | tail -12
echo "== python bench.py (defaults), wall time"
SECONDS=0
timeout 900 python bench.py > gpurun_out/r03_bench_check3.json 2> gpurun_out/r03_bench_check3.err
echo "bench.py wall: ${SECONDS} s (fresh process; the first import of torch on this box happened in the pytest run above)"
cut -c1-600 gpurun_out/r03_bench_check3.json
} > gpurun_out/r03_check3.log 2>&1
tail -c 4000 gpurun_out/r03_check3.log
