#!/bin/bash
# round 4, call F: dense phases with the split dense GEMM; rocprofv3 kernel trace + matrix-core busy of the dense phases at batch 64
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "gemm_256" 2>&1 | grep -E "^\[gemm256|passed|failed|^FAILED|^E  " | tee gpurun_out/r04f_gemm256.txt | tail -20
timeout 600 python scripts/prof_dense.py --batches 16,64 --iters 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04f_dense.txt | tail -12
cd /tmp; rm -rf /tmp/profd
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/profd -o d --output-format csv -- python $R/scripts/prof_dense.py --batches 64 --iters 2 > $R/gpurun_out/r04f_prof_dense.log 2>&1
for f in $(find /tmp/profd -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r04f_dense_b64_kernel_stats.csv; done
head -24 $R/gpurun_out/r04f_dense_b64_kernel_stats.csv | cut -c1-200
rm -rf /tmp/pmc_mfma
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_mfma -o p --output-format csv -- python $R/scripts/prof_dense.py --batches 64 --iters 1 > $R/gpurun_out/r04f_pmc_mfma.log 2>&1
python $R/scripts/pmc_summary.py $R/gpurun_out/r04_pmc_dense_mfma_raw.json /tmp/pmc_mfma > $R/gpurun_out/r04f_pmc_mfma_summary.log 2>&1
tail -5 $R/gpurun_out/r04f_pmc_mfma_summary.log
ls /tmp/pmc_mfma | head
