#!/bin/bash
# round 5, call E: the two-launch 8-row layer after the fixes (single rounding of the attention output, cache requests behind the MFMAs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
timeout 600 python -m pytest tests/test_gpu_rows_attn.py -q -s -p no:cacheprovider > $O/rows_attn_tests.txt 2>&1; echo "rows_attn rc $?"
grep -E "^\[8 rows|passed|failed|^E  " $O/rows_attn_tests.txt | cut -c1-600 | tail -16
for early in 2 1 0; do
timeout 200 python scripts/trace_step.py --batch 8 --lens 3858 --options rows_attn_early=$early > $O/timeline_b8_early$early.txt 2>&1
grep -v amdgpu.ids $O/timeline_b8_early$early.txt
done
timeout 200 python scripts/trace_step.py --batch 8 --lens 300,7300 > $O/timeline_b8_two_launches.txt 2>&1
grep -v amdgpu.ids $O/timeline_b8_two_launches.txt
