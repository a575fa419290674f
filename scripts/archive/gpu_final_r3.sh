#!/bin/bash
# Round-3 evidence in one gpurun call: bench.py with its defaults, the rocprofv3 kernel trace of the bench command, PMC passes (HBM traffic of
# the decode launches; matrix-core busy of the dense phases), BASELINE configs 3 / 5 and the batch-8 per-rank workload as full generations,
# the 1-rank torchrun bench (RCCL broadcast path), the in-kernel step timeline.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== bench (defaults)"
timeout 900 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -c 3000 gpurun_out/r03_bench.json; tail -c 400 gpurun_out/r03_bench.err
echo "== rocprof kernel trace of the bench command"
cd /tmp; rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r3 --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched-table > $R/gpurun_out/r03_prof.log 2>&1
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r03_bench_kernel_stats.csv; done
head -12 $R/gpurun_out/r03_bench_kernel_stats.csv | cut -c1-200
echo "== PMC passes"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p --output-format csv -- python $R/scripts/prof_step.py --options "use_graph=0" --steps 2 --gen 96 > $R/gpurun_out/pmc_$C.log 2>&1
done
python $R/scripts/pmc_summary.py $R/gpurun_out/r03_pmc_decode_raw.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $R/gpurun_out/pmc_decode_summary.log 2>&1
rm -rf /tmp/pmc_mfma
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_mfma -o p --output-format csv -- python $R/scripts/prof_dense.py --batches 64 --iters 1 > $R/gpurun_out/pmc_mfma.log 2>&1
python $R/scripts/pmc_summary.py $R/gpurun_out/r03_pmc_dense_mfma_raw.json /tmp/pmc_mfma > $R/gpurun_out/pmc_mfma_summary.log 2>&1
python $R/scripts/pmc_r2_report.py $R/gpurun_out/r03_pmc_decode_raw.json $R/gpurun_out/r03_pmc_dense_mfma_raw.json $R/gpurun_out r03
head -c 1500 $R/gpurun_out/r03_pmc_decode_traffic.json; head -c 1800 $R/gpurun_out/r03_pmc_dense_mfma.json
cd $R
echo "== configs 3, 5 and the batch-8 per-rank workload"
timeout 600 python bench.py --batch 64 --sampling --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03_cfg3_b64_sampling.json 2> gpurun_out/r03_cfg3.err; cut -c1-400 gpurun_out/r03_cfg3_b64_sampling.json
timeout 600 python bench.py --batch 8 --faces 1600 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03_cfg5_b8_1600.json 2> gpurun_out/r03_cfg5.err; cut -c1-400 gpurun_out/r03_cfg5_b8_1600.json
timeout 600 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_b8_800.json 2> gpurun_out/r03_b8.err; cut -c1-400 gpurun_out/r03_b8_800.json
echo "== torchrun, 1 rank (RCCL arena broadcast path)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-batched-table > gpurun_out/r03_torchrun1.json 2> gpurun_out/r03_torchrun1.err
cut -c1-400 gpurun_out/r03_torchrun1.json
echo "== step timeline (batch 1)"
timeout 300 python scripts/trace_step.py --lens 300,3800,7400 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_trace_b1.log; cat gpurun_out/r03_trace_b1.log
