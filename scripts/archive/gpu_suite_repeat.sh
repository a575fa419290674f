#!/bin/bash
# the whole GPU suite N times in one box (flake hunt): every failure message kept
mkdir -p gpurun_out; export TMPDIR=/tmp
N=${1:-3}
: > gpurun_out/r04_suite_repeat.txt
for i in $(seq 1 $N); do
  timeout 1200 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "amdgpu.ids" > /tmp/suite_$i.txt
  echo "run $i: $(grep -E 'passed|failed' /tmp/suite_$i.txt | tail -1)" >> gpurun_out/r04_suite_repeat.txt
  if grep -qE "failed|error" /tmp/suite_$i.txt; then grep -E "^E  |^FAILED|^ERROR" /tmp/suite_$i.txt | cut -c1-400 | head -40 >> gpurun_out/r04_suite_repeat.txt; cp /tmp/suite_$i.txt gpurun_out/r04_suite_fail_$i.txt; fi
done
cat gpurun_out/r04_suite_repeat.txt | cut -c1-300
