#!/bin/bash
# round 5, call G: deep-cache parity tests (full-length reference anchor), then the whole GPU suite as the driver runs it, then smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_long_context.py -q -s -p no:cacheprovider > $O/long_context.txt 2>&1; echo "long context rc $?"
grep -E "^\[|passed|failed|^E  " $O/long_context.txt | cut -c1-900 | tail -24
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/suite.txt 2>&1; echo "suite rc $?"
tail -5 $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.txt
