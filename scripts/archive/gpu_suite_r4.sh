#!/bin/bash
# the driver's round-end sequence on a fresh box: the GPU suite (untasksetted), smoke(); full output kept, failures summarised
mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
timeout 2400 python -m pytest tests/ -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04_suite_final.txt
echo "suite wall $(( $(date +%s) - T0 )) s"
grep -E "passed|failed" gpurun_out/r04_suite_final.txt | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r04_suite_final.txt | head -20
grep -E "^E  " gpurun_out/r04_suite_final.txt | cut -c1-300 | head -30
timeout 300 python -c "import __graft_entry__ as g; import time; t=time.time(); g.smoke(); print('smoke ok', round(time.time()-t,1), 's')" 2>&1 | grep -v amdgpu.ids | tail -3
