#!/bin/bash
# Round-2 PMC evidence (separate rocprofv3 --pmc passes, --kernel-trace only, as the MI355X guide prescribes):
#   1. HBM traffic of the decode-step launches (FETCH_SIZE, WRITE_SIZE) with the shipped defaults, graph off (rocprofv3 aborts on the replayed graph)
#   2. matrix-core busy cycles of the dense phases at batch 64 (SQ_VALU_MFMA_BUSY_CYCLES vs GRBM_GUI_ACTIVE)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p --output-format csv -- python $R/scripts/prof_step.py --options "use_graph=0" --steps 2 --gen 96 > $R/gpurun_out/pmc_$C.log 2>&1
  tail -1 $R/gpurun_out/pmc_$C.log
done
python $R/scripts/pmc_summary.py $R/gpurun_out/r02_pmc_decode_raw.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $R/gpurun_out/pmc_decode_summary.log 2>&1
rm -rf /tmp/pmc_mfma
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_mfma -o p --output-format csv -- python $R/scripts/prof_dense.py --batches 64 --iters 1 > $R/gpurun_out/pmc_mfma.log 2>&1
tail -2 $R/gpurun_out/pmc_mfma.log
python $R/scripts/pmc_summary.py $R/gpurun_out/r02_pmc_dense_mfma_raw.json /tmp/pmc_mfma > $R/gpurun_out/pmc_mfma_summary.log 2>&1
python $R/scripts/pmc_r2_report.py $R/gpurun_out/r02_pmc_decode_raw.json $R/gpurun_out/r02_pmc_dense_mfma_raw.json $R/gpurun_out
cat $R/gpurun_out/r02_pmc_decode_traffic.json | head -40
cat $R/gpurun_out/r02_pmc_dense_mfma.json | head -40
