#!/bin/bash
# Round-5 evidence in one gpurun call: bench.py with its defaults, the rocprofv3 kernel trace of the bench command and of the batch-8 per-rank
# workload, PMC passes (HBM traffic of the decode launches at batch 1 and at 8 rows), BASELINE configs 3 / 5 and the batch-8 workload, the
# 1-rank torchrun bench, the gemm256 stress on this (second) box, the in-kernel step timelines.
mkdir -p gpurun_out/r5final; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5final
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
echo "== bench (defaults)"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "phases_ms", "dense_phases", "encoder_max_abs_err", "tokens_distinct", "fused_launch_health", "measured_peaks", "fp32_exact", "fp16_policy", "batched_decode_steps", "cpu_baseline"):
    print(k, "=", json.dumps(d.get(k)))
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if k != "classes"}))
PY
tail -c 300 $O/bench.err
echo "== rocprof kernel trace of the bench command"
cd /tmp; rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r5 --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched-table > $O/prof_bench.json 2> $O/prof.log
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $O/bench_kernel_stats.csv; done
head -8 $O/bench_kernel_stats.csv | cut -c1-200
echo "== kernel stats of the batch-8 bench"
rm -rf /tmp/prof8
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof8 -o r5 --output-format csv -- python $R/bench.py --batch 8 --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_b8.json 2> $O/prof_b8.log
for f in $(find /tmp/prof8 -name "*kernel_stats*.csv"); do cp $f $O/bench_batch8_kernel_stats.csv; done
head -8 $O/bench_batch8_kernel_stats.csv | cut -c1-200
echo "== PMC passes: decode traffic, batch 1 and 8 rows"
for B in 1 8; do
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p --output-format csv -- python $R/scripts/prof_step.py --batch $B --options "use_graph=0" --steps 2 --gen 256 --no-profile > $O/pmc_${C}_b$B.log 2>&1
done
python $R/scripts/pmc_summary.py $O/pmc_decode_raw_b$B.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_decode_summary_b$B.log 2>&1
echo '{}' > $O/empty.json
python $R/scripts/pmc_r2_report.py $O/pmc_decode_raw_b$B.json $O/empty.json $O r05_b$B
head -c 1500 $O/r05_b${B}_pmc_decode_traffic.json; echo
done
cd $R
echo "== configs 3, 5 and the batch-8 per-rank workload"
timeout 600 python bench.py --batch 64 --sampling --steps 1 --warmup 0 --no-cpu-baseline > $O/cfg3_b64_sampling.json 2> $O/cfg3.err; cut -c1-300 $O/cfg3_b64_sampling.json; echo
timeout 600 python bench.py --batch 8 --faces 1600 --steps 1 --warmup 0 --no-cpu-baseline > $O/cfg5_b8_1600.json 2> $O/cfg5.err; cut -c1-300 $O/cfg5_b8_1600.json; echo
timeout 600 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > $O/b8_800.json 2> $O/b8.err; cut -c1-300 $O/b8_800.json; echo
echo "== torchrun, 1 rank (RCCL arena broadcast path)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-batched-table > $O/torchrun1.json 2> $O/torchrun1.err
cut -c1-300 $O/torchrun1.json; echo
echo "== gemm256 stress on this box (release build, then the debug build)"
timeout 300 python scripts/stress_gemm256.py 600 > $O/stress_gemm256_box2.txt 2>&1; tail -2 $O/stress_gemm256_box2.txt
MA_DEBUG=1 timeout 600 python scripts/stress_gemm256.py 100 > $O/stress_gemm256_debug.txt 2>&1; tail -2 $O/stress_gemm256_debug.txt
echo "== step timelines"
timeout 300 python scripts/trace_step.py --batch 8 --lens 300,3858,7300 --dist 2>&1 | grep -v amdgpu.ids > $O/trace_b8.txt; tail -30 $O/trace_b8.txt
timeout 300 python scripts/trace_step.py --lens 300,3800,7400 2>&1 | grep -v amdgpu.ids > $O/trace_b1.txt; tail -24 $O/trace_b1.txt
