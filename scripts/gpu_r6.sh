#!/bin/bash
# Round 6, GPU box: the stages named in $STAGES (space separated) on the tree as it stands; everything lands in gpurun_out/$TAG/
#   suite   pytest -m gpu + smoke            dense   scripts/prof_dense.py (dense-phase times per batch) + its rocprof kernel stats at B = $PB
#   calib   scripts/calib_library_gemm.py    bench   bench.py (defaults)         b8 / cfg5   bench.py --batch 8 [--faces 1600]
#   exp     the MA_EXPERIMENTAL tests        trace8  the 8-row step's timeline
#   gemmtests  the dense GEMM kernel tests (-s: their timing lines)    stress  scripts/stress_gemm256.py
#   fp32    the fp32 policy on the fused launches (bitwise tests, gates, one mesh each way)      cfg3    bench.py --batch 64 --sampling
#   soak    the 8-row two-launch layer in round 5's failing form      pmc     HBM bytes of the decode launches + matrix-core busy of the dense phases (separate passes)
#   phase   kernel trace of ONE dense phase      benchprof   rocprofv3 --kernel-trace --stats of bench.py and of bench.py --batch 8
# other one-off measurements of the round: scripts/time_fp32_policy.py, provoke_queue_eviction.py, soak_fused_engine_churn.py, analyze_tail_chain.py (see profiles/README.md)
TAG=${TAG:-r6}; STAGES=${STAGES:-suite}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
for st in $STAGES; do
  echo "== $st"
  case $st in
    suite)
      timeout ${SUITE_TMO:-1200} python -m pytest tests -m gpu -x -q -p no:cacheprovider ${PYTEST_ARGS} 2>&1 | grep -v amdgpu.ids > $O/suite.txt; tail -6 $O/suite.txt
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt ;;
    exp)
      MA_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_rows_fused.py tests/test_gpu_rows_attn.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids > $O/suite_experimental.txt; tail -3 $O/suite_experimental.txt
      MA_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -p no:cacheprovider -k "layernorm_finished or kv_written or split_along_k" 2>&1 | grep -v amdgpu.ids > $O/suite_experimental_prefill.txt; tail -3 $O/suite_experimental_prefill.txt ;;
    fp32)
      # the fp32 ("exact") policy on the fused batch-1 chain: bitwise against the five-launch chain, the long-context / reference-anchor gates, one timed mesh each way
      timeout 900 python -m pytest tests/test_gpu_persist.py -x -q -p no:cacheprovider -s -k "fp32 and (fused or fc2)" 2>&1 | grep -v amdgpu.ids > $O/fp32_fused_bitwise.txt; grep -E "A/B|passed|failed|Error|assert" $O/fp32_fused_bitwise.txt | cut -c1-300 | tail -20
      timeout 900 python -m pytest tests/test_gpu_long_context.py tests/test_gpu_reference_anchor.py -x -q -p no:cacheprovider -k "fp32" 2>&1 | grep -v amdgpu.ids > $O/fp32_gates.txt; tail -4 $O/fp32_gates.txt
      timeout 600 python scripts/time_fp32_policy.py 2>&1 | grep -v amdgpu.ids > $O/fp32_policy_mesh.txt; cat $O/fp32_policy_mesh.txt ;;
    dense)
      timeout 600 python scripts/prof_dense.py --batches ${BATCHES:-16,64} 2>&1 | grep -v amdgpu.ids > $O/dense.txt; cat $O/dense.txt
      cd /tmp; rm -rf /tmp/prof_d
      timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o d --output-format csv -- python $R/scripts/prof_dense.py --batches ${PB:-64} --iters 2 > $O/prof_dense_rocprof.log 2>&1
      for f in $(find /tmp/prof_d -name "*kernel_stats*.csv"); do cp $f $O/dense_b${PB:-64}_kernel_stats.csv; done
      head -14 $O/dense_b${PB:-64}_kernel_stats.csv | cut -c1-180; cd $R ;;
    calib)
      timeout 600 python scripts/calib_library_gemm.py 2>&1 | grep -v amdgpu.ids > $O/calib_library_gemm.txt; cat $O/calib_library_gemm.txt ;;
    bench)
      timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json; echo ;;
    b8)
      timeout 600 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > $O/b8_800.json 2> $O/b8.err; cut -c1-400 $O/b8_800.json; echo ;;
    cfg5)
      timeout 600 python bench.py --batch 8 --faces 1600 --steps 1 --warmup 0 --no-cpu-baseline > $O/cfg5_b8_1600.json 2> $O/cfg5.err; cut -c1-400 $O/cfg5_b8_1600.json; echo ;;
    trace8)
      timeout 300 python scripts/trace_step.py --batch 8 --lens 300,3858,7300 --dist 2>&1 | grep -v amdgpu.ids > $O/trace_b8.txt; grep -E "len|attn |fc1 " $O/trace_b8.txt ;;
    gemmtests)
      timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "gemm" -s 2>&1 | grep -v amdgpu.ids > $O/gemmtests.txt; grep -E "gemm256|passed|failed|Error" $O/gemmtests.txt | cut -c1-330 | tail -40 ;;
    stress)
      timeout 900 python scripts/stress_gemm256.py ${STRESS_N:-600} 2>&1 | grep -v amdgpu.ids > $O/stress_gemm256.txt; tail -30 $O/stress_gemm256.txt ;;
    soak)
      # the form in which round 5's one fall-back was seen, then the vector-sweep control
      timeout 1500 python scripts/stress_rows_b8.py ${SOAK_N:-30} 1600 forced 6 2>&1 | grep -v amdgpu.ids > $O/soak_rows_b8_forced_early6.txt; tail -4 $O/soak_rows_b8_forced_early6.txt
      timeout 1500 python scripts/stress_rows_b8.py ${SOAK_N:-30} 1600 forced 3 2>&1 | grep -v amdgpu.ids > $O/soak_rows_b8_forced_early3.txt; tail -4 $O/soak_rows_b8_forced_early3.txt ;;
    pmc)
      # PMC passes (separate runs, --kernel-trace only, as the MI355X guide prescribes): HBM bytes of the decode launches, matrix-core busy of the dense phases
      cd /tmp
      for C in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_$C
        timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p --output-format csv -- python $R/scripts/prof_step.py --options "use_graph=0" --steps 2 --gen 96 > $O/pmc_$C.log 2>&1
      done
      python $R/scripts/pmc_summary.py $O/r06_pmc_decode_raw.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_decode_summary.log 2>&1
      rm -rf /tmp/pmc_mfma
      timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_mfma -o p --output-format csv -- python $R/scripts/prof_dense.py --batches 64 --iters 1 --options prefill_tail=0 > $O/pmc_mfma.log 2>&1      # (one-stream prefill: counter collection runs kernels one at a time; the tail chain's 10 small launches per layer would only add their collection overhead to the denominator)
      python $R/scripts/pmc_summary.py $O/r06_pmc_dense_mfma_raw.json /tmp/pmc_mfma > $O/pmc_mfma_summary.log 2>&1
      python $R/scripts/pmc_r2_report.py $O/r06_pmc_decode_raw.json $O/r06_pmc_dense_mfma_raw.json $O r06
      head -c 600 $O/r06_pmc_decode_traffic.json; echo; python - <<PY
import json
d = json.load(open("$O/r06_pmc_dense_mfma.json"))
print({k: v for k, v in d.items() if k != "per_kernel" and k != "formula" and k != "source"})
for k, v in sorted(d["per_kernel"].items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"] * kv[1]["dispatches"])[:14]:
    print(f"{v['mfma_busy_frac']:.3f} x{v['dispatches']:4d}  {k[:110]}")
PY
      cd $R ;;
    phase)
      # kernel trace of ONE dense phase ($PHASE = prefill | detok) at $PB samples
      cd /tmp; rm -rf /tmp/prof_p
      timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o d --output-format csv -- python $R/scripts/prof_dense.py --batches ${PB:-64} --iters 3 --phases ${PHASE:-detok} > $O/prof_${PHASE:-detok}_rocprof.log 2>&1
      for f in $(find /tmp/prof_p -name "*kernel_stats*.csv"); do cp $f $O/${PHASE:-detok}_b${PB:-64}_kernel_stats.csv; done
      grep dense $O/prof_${PHASE:-detok}_rocprof.log; head -24 $O/${PHASE:-detok}_b${PB:-64}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-170; cd $R ;;
    benchprof)
      # the kernel trace of the bench command itself (roofline.avg_launch_us must agree with it) and of the 8-row workload
      cd /tmp; rm -rf /tmp/prof_b /tmp/prof_b8
      timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batched-table > $O/bench_profiled.json 2> $O/bench_profiled.err
      for f in $(find /tmp/prof_b -name "*kernel_stats*.csv"); do cp $f $O/bench_kernel_stats.csv; done
      head -6 $O/bench_kernel_stats.csv | cut -c1-200
      timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b8 -o b --output-format csv -- python $R/bench.py --batch 8 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_b8_profiled.json 2> $O/bench_b8_profiled.err
      for f in $(find /tmp/prof_b8 -name "*kernel_stats*.csv"); do cp $f $O/bench_batch8_kernel_stats.csv; done
      head -5 $O/bench_batch8_kernel_stats.csv | cut -c1-200; cd $R ;;
    cfg3)
      timeout 900 python bench.py --batch 64 --sampling --steps 1 --warmup 0 --no-cpu-baseline > $O/cfg3_b64_sampling.json 2> $O/cfg3.err; cut -c1-400 $O/cfg3_b64_sampling.json; echo ;;
    *) echo "unknown stage $st" ;;
  esac
done
