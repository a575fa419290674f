#!/bin/bash
# The driver's GPU tier (pytest -m gpu, smoke) on the tree as it stands, then the tests of the measured-and-rejected launches against the
# MA_EXPERIMENTAL build of the same sources.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5suite; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids > $O/suite.txt; tail -5 $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
MA_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_persist.py tests/test_gpu_rows_fused.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids > $O/suite_experimental.txt; tail -3 $O/suite_experimental.txt
