#!/bin/bash
# round 5, call C: the two-launch 8-row layer (bitwise vs five launches, launch counts, A/B), timeline of the default form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
timeout 600 python -m pytest tests/test_gpu_rows_attn.py -q -s -p no:cacheprovider > $O/rows_attn_tests.txt 2>&1; echo "rows_attn rc $?"
grep -E "^\[8 rows|passed|failed|Error|rror:" $O/rows_attn_tests.txt | cut -c1-700 | tail -20
timeout 200 python scripts/trace_step.py --batch 8 --lens 300,3858,7300 > $O/timeline_b8_two_launches.txt 2>&1
grep -v amdgpu.ids $O/timeline_b8_two_launches.txt
