#!/bin/bash
# The driver's round-end sequence on one box: `pytest tests -x -q -m gpu` (under taskset -c 0-7: a box whose cgroup grants few
# cores must not change the result), then smoke(), each timed.  Output -> gpurun_out/suite.log
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== nproc $(nproc)"; date
  SECONDS=0
  taskset -c 0-7 timeout ${SUITE_TIMEOUT:-1100} python -m pytest tests/ -x -q -m gpu 2>&1 | tail -80
  echo "== pytest -m gpu wall: ${SECONDS} s (taskset -c 0-7, fresh box, includes the first import of torch)"
  echo "== smoke"; ( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -6
} > gpurun_out/suite.log 2>&1
tail -c 6000 gpurun_out/suite.log
