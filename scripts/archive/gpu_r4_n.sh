#!/bin/bash
# round 4, call N: dense-phase row kernels with their requests up front (LayerNorm rows, 16-byte KV fill), two-polynomial erf in the GELU
# epilogue, 8 chunks in flight in the 33..64-row skinny GEMM: tests, dense phases with kernel stats, batched steps A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_reference_anchor.py -q -m gpu 2>&1 | grep -v amdgpu.ids > gpurun_out/r04n_tests.txt; tail -4 gpurun_out/r04n_tests.txt; grep -E "^E  |^FAILED" gpurun_out/r04n_tests.txt | head
echo "== dense phases"
timeout 300 python scripts/prof_dense.py --batches 16,64 --iters 3 2>&1 | grep dense
cd /tmp; rm -rf /tmp/profd
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profd -o d --output-format csv -- python $R/scripts/prof_dense.py --batches 64 --iters 2 > $R/gpurun_out/r04n_prof_dense.log 2>&1
for f in $(find /tmp/profd -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r04n_dense_b64_kernel_stats.csv; done
head -24 $R/gpurun_out/r04n_dense_b64_kernel_stats.csv | cut -c1-150
cd $R
echo "== steps"
for B in 24 64; do timeout 300 python scripts/prof_step.py --batch $B --steps 4 --options "mfma_chunks=4;mfma_chunks=8" 2>&1 | grep "len" ; done
