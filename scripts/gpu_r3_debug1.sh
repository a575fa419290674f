#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== cProfile of the oracle cross-check test under pytest"
timeout 500 python -m cProfile -o /tmp/p.out -m pytest tests/test_gpu_oracle_device.py -x -q -k fp32 2>&1 | tail -5
python -c "import pstats; pstats.Stats('/tmp/p.out').sort_stats('tottime').print_stats(22)" 2>&1 | tail -40
echo "== fallback test"
timeout 300 python -m pytest tests/test_gpu_persist.py -x -q -s -k "fall_back" 2>&1 | tail -15
echo "== reference anchor tests"
timeout 600 python -m pytest tests/test_gpu_reference_anchor.py -x -q -s 2>&1 | tail -30
} > gpurun_out/r03_debug1.log 2>&1
tail -c 9000 gpurun_out/r03_debug1.log
