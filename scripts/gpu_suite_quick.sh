#!/bin/bash
# whole GPU suite without the slow fidelity reports, then a short bench for the phase times
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  SECONDS=0
  taskset -c 0-7 timeout 1100 python -m pytest tests/ -x -q -m gpu -k "not fidelity" 2>&1 | tail -40
  echo "== wall ${SECONDS} s"
  timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batched-table 2>gpurun_out/bench_quick.err | tee gpurun_out/bench_quick.json | cut -c1-1500
  tail -3 gpurun_out/bench_quick.err
} > gpurun_out/suite_quick.log 2>&1
tail -c 7000 gpurun_out/suite_quick.log
