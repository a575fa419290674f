#!/bin/bash
# round 4, call V: LayerNorm rows kernel with four rows per wave: tests, dense phases
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 | grep -v amdgpu.ids > gpurun_out/r04v_kernels.txt; grep -E "passed|failed|^FAILED|^E  " gpurun_out/r04v_kernels.txt | cut -c1-300 | head
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_reference_anchor.py -q -m gpu 2>&1 | grep -v amdgpu.ids > gpurun_out/r04v_tests.txt; grep -E "passed|failed|^FAILED|^E  " gpurun_out/r04v_tests.txt | cut -c1-300 | head
timeout 300 python scripts/prof_dense.py --batches 16,64 --iters 3 2>&1 | grep dense
cd /tmp; rm -rf /tmp/profd
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profd -o d --output-format csv -- python $R/scripts/prof_dense.py --batches 64 --iters 2 > $R/gpurun_out/r04v_prof_dense.log 2>&1
for f in $(find /tmp/profd -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r04v_dense_b64_kernel_stats.csv; done
grep -E "ln_rows2|kv_fill2|vt_pack|codes_gather|attention_mfma2|gemm_dec" $R/gpurun_out/r04v_dense_b64_kernel_stats.csv | sed 's/(float const.*)",/",/; s/(unsigned short const.*)",/",/; s/(long long const.*)",/",/; s/(ma::.*Args)",/",/' | cut -c1-160
