#!/bin/bash
# The driver's round-end sequence on one box, EXACTLY as the driver runs it (no taskset): `python -m pytest tests/ -x -q -m gpu`,
# then smoke(), each timed.  SUITE_PIN="0" / "0-1" additionally repeats the suite under `taskset -c $SUITE_PIN`.
# Output -> gpurun_out/r03_suite*.log
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # $1 = log tag, rest = command prefix
  tag=$1; shift
  {
    echo "== nproc $(nproc) loadavg $(cat /proc/loadavg) cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) prefix: ${*:-none}"; date
    SECONDS=0
    "$@" timeout ${SUITE_TIMEOUT:-1100} python -m pytest tests/ -x -q -m gpu 2>&1 | tail -${SUITE_TAIL:-400}
    echo "== pytest -m gpu wall: ${SECONDS} s (prefix: ${*:-none}; fresh box for the first run: includes the first import of torch)"
  } > gpurun_out/r03_suite_$tag.log 2>&1
  tail -c 1500 gpurun_out/r03_suite_$tag.log
}
run untasksetted
{ echo "== smoke"; ( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -6; } >> gpurun_out/r03_suite_untasksetted.log 2>&1
tail -8 gpurun_out/r03_suite_untasksetted.log
for pin in $SUITE_PIN; do run taskset_$pin taskset -c $pin; done
