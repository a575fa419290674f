#!/bin/bash
# BASELINE.json configs 3 and 5 as full generations (one pass each, no warm-up pass beyond graph capture)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --batch 64 --sampling --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_cfg3_b64_sampling.json 2> gpurun_out/bench_cfg3.err
tail -c 1800 gpurun_out/bench_cfg3_b64_sampling.json; tail -3 gpurun_out/bench_cfg3.err
timeout 900 python bench.py --batch 8 --faces 1600 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_cfg5_b8_1600.json 2> gpurun_out/bench_cfg5.err
tail -c 1800 gpurun_out/bench_cfg5_b8_1600.json; tail -3 gpurun_out/bench_cfg5.err
