#!/bin/bash
# Round-4 evidence in one gpurun call: bench.py with its defaults, the rocprofv3 kernel trace of the bench command (fused-launch health of the
# PROFILED run kept beside it), PMC passes (HBM traffic of the decode launches; matrix-core busy of the dense phases), the GPU tests of the
# MA_EXPERIMENTAL build, BASELINE configs 3 / 5 and the batch-8 per-rank workload, the 1-rank torchrun bench, the in-kernel step timeline.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== bench (defaults)"
timeout 900 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench.json"))
for k in ("value", "ms_per_step", "phases_ms", "dense_phases", "encoder_max_abs_err", "tokens_distinct", "fused_launch_health", "measured_peaks", "fp32_exact", "fp16_policy", "batched_decode_steps", "cpu_baseline"):
    print(k, "=", json.dumps(d.get(k)))
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if k != "classes"}))
print({k: v["avg_launch_us"] for k, v in d["roofline"]["classes"].items()})
PY
tail -c 300 gpurun_out/r04_bench.err
echo "== rocprof kernel trace of the bench command"
cd /tmp; rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r4 --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched-table > $R/gpurun_out/r04_prof_bench.json 2> $R/gpurun_out/r04_prof.log
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r04_bench_kernel_stats.csv; done
for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do python - $f $R/gpurun_out/r04_long_dispatches.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w")
durs = {}
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    durs.setdefault(r["Kernel_Name"].split("(")[0][:60], []).append((d, int(r["Start_Timestamp"])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
print("decode-launch dispatches longer than 1 ms in the profiled bench run (us, time since first dispatch):", file=out)
for k, v in durs.items():
    if "oproj_fc1" in k or "qkv_attn" in k or "gemv_kernel" in k:
        v.sort()
        med = v[len(v) // 2][0]
        lng = [(round(d, 1), round((t - t0) / 1e9, 3)) for d, t in v if d > 1000.0]
        print(f"{k}: n={len(v)} median={med:.2f} max={v[-1][0]:.1f} long={lng}", file=out)
PY
done
cat $R/gpurun_out/r04_long_dispatches.txt
python - <<PY
import json
d = json.load(open("$R/gpurun_out/r04_prof_bench.json"))
print("health of the PROFILED run:", json.dumps(d.get("fused_launch_health")))
open("$R/gpurun_out/r04_long_dispatches.txt", "a").write("fused_launch_health of the same (profiled) run: " + json.dumps(d.get("fused_launch_health")) + "\n")
PY
head -14 $R/gpurun_out/r04_bench_kernel_stats.csv | cut -c1-220
echo "== PMC passes"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p --output-format csv -- python $R/scripts/prof_step.py --options "use_graph=0" --steps 2 --gen 96 > $R/gpurun_out/pmc_$C.log 2>&1
done
python $R/scripts/pmc_summary.py $R/gpurun_out/r04_pmc_decode_raw.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $R/gpurun_out/pmc_decode_summary.log 2>&1
rm -rf /tmp/pmc_mfma
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_mfma -o p --output-format csv -- python $R/scripts/prof_dense.py --batches 64 --iters 1 > $R/gpurun_out/pmc_mfma.log 2>&1
python $R/scripts/pmc_summary.py $R/gpurun_out/r04_pmc_dense_mfma_raw.json /tmp/pmc_mfma > $R/gpurun_out/pmc_mfma_summary.log 2>&1
python $R/scripts/pmc_r2_report.py $R/gpurun_out/r04_pmc_decode_raw.json $R/gpurun_out/r04_pmc_dense_mfma_raw.json $R/gpurun_out r04
head -c 1200 $R/gpurun_out/r04_pmc_decode_traffic.json; echo; head -c 2500 $R/gpurun_out/r04_pmc_dense_mfma.json; echo
cd $R
echo "== GPU tests of the MA_EXPERIMENTAL build (persistent step, rows-looped layer, layer-pair launch, GEMM A/B variants)"
MA_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_rows_fused.py tests/test_gpu_kernels.py -q -m gpu 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04_suite_experimental.txt
tail -4 gpurun_out/r04_suite_experimental.txt; grep -E "^E  |^FAILED" gpurun_out/r04_suite_experimental.txt | head
echo "== configs 3, 5 and the batch-8 per-rank workload"
timeout 600 python bench.py --batch 64 --sampling --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r04_cfg3_b64_sampling.json 2> gpurun_out/r04_cfg3.err; cut -c1-300 gpurun_out/r04_cfg3_b64_sampling.json; echo
timeout 600 python bench.py --batch 8 --faces 1600 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r04_cfg5_b8_1600.json 2> gpurun_out/r04_cfg5.err; cut -c1-300 gpurun_out/r04_cfg5_b8_1600.json; echo
timeout 600 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_b8_800.json 2> gpurun_out/r04_b8.err; cut -c1-300 gpurun_out/r04_b8_800.json; echo
echo "== torchrun, 1 rank (RCCL arena broadcast path)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-batched-table > gpurun_out/r04_torchrun1.json 2> gpurun_out/r04_torchrun1.err
cut -c1-300 gpurun_out/r04_torchrun1.json; echo
echo "== kernel stats of the batch-8 bench"
cd /tmp; rm -rf /tmp/prof8
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof8 -o r4 --output-format csv -- python $R/bench.py --batch 8 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r04_prof_b8.json 2> $R/gpurun_out/r04_prof_b8.log
for f in $(find /tmp/prof8 -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r04_bench_batch8_kernel_stats.csv; done
head -8 $R/gpurun_out/r04_bench_batch8_kernel_stats.csv | cut -c1-200
cd $R
echo "== dense phases"
timeout 300 python scripts/prof_dense.py --batches 16,64 --iters 3 2>&1 | grep dense | tee gpurun_out/r04_dense_phases_final.txt
echo "== step timeline (batch 8, then batch 1)"
timeout 300 python scripts/trace_step.py --batch 8 --lens 300,3858 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_trace_b8_final.log; tail -16 gpurun_out/r04_trace_b8_final.log
timeout 300 python scripts/trace_step.py --lens 300,3800,7400 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_trace_b1.log; tail -40 gpurun_out/r04_trace_b1.log
