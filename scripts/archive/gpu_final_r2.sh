#!/bin/bash
# Round-2 evidence in one gpurun call: the driver's sequence (pytest -m gpu, smoke), bench.py with its defaults, the rocprofv3 kernel
# trace of the bench command, BASELINE configs 3 and 5 as full generations, the in-kernel step timeline, the 1-rank torchrun bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
{
  echo "== device"; python -c "import torch;print(torch.cuda.get_device_name(0), torch.cuda.device_count())"; echo "nproc $(nproc)"; date
  SECONDS=0
  taskset -c 0-7 timeout 1100 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -60
  echo "== pytest -m gpu wall: ${SECONDS} s (taskset -c 0-7, fresh box, includes the first import of torch)"
  echo "== smoke"; ( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -6
} > gpurun_out/final_suite.log 2>&1
tail -c 2500 gpurun_out/final_suite.log
echo "== bench (defaults)"
timeout 1200 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -c 2500 gpurun_out/final_bench.json; tail -c 600 gpurun_out/final_bench.err
echo "== rocprof kernel trace of the bench command"
cd /tmp; rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r2 --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched-table > $R/gpurun_out/final_prof.log 2>&1
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/final_bench_kernel_stats.csv; done
head -14 $R/gpurun_out/final_bench_kernel_stats.csv | cut -c1-220
cd $R
echo "== configs 3 and 5"
timeout 900 python bench.py --batch 64 --sampling --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final_cfg3_b64_sampling.json 2> gpurun_out/final_cfg3.err
cut -c1-700 gpurun_out/final_cfg3_b64_sampling.json; tail -2 gpurun_out/final_cfg3.err
timeout 900 python bench.py --batch 8 --faces 1600 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final_cfg5_b8_1600.json 2> gpurun_out/final_cfg5.err
cut -c1-700 gpurun_out/final_cfg5_b8_1600.json; tail -2 gpurun_out/final_cfg5.err
timeout 900 python bench.py --batch 8 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final_cfg4_b8_800.json 2> gpurun_out/final_cfg4.err
cut -c1-700 gpurun_out/final_cfg4_b8_800.json; tail -2 gpurun_out/final_cfg4.err
echo "== step timeline"
timeout 600 python scripts/trace_step.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final_trace.log; tail -30 gpurun_out/final_trace.log
echo "== torchrun, 1 rank (RCCL arena broadcast path)"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-batched-table > gpurun_out/final_torchrun1.json 2> gpurun_out/final_torchrun1.err
cut -c1-600 gpurun_out/final_torchrun1.json; grep -i "broadcast\|rccl\|nccl" gpurun_out/final_torchrun1.err | tail -5
