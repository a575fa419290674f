#!/bin/bash
# round 5, call A: the fused 8-row first half (bitwise vs the three launches, launch count, A/B, timeline), the deep-cache kernel tests,
# the gemm256 stress, the ASan attempt, and -- on the host cores meanwhile -- the full CPU baseline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
(python scripts/cpu_baseline_full.py 16 > $O/cpu_baseline_full.json 2> $O/cpu_baseline_full.log) &
CPU_PID=$!
timeout 420 python -m pytest tests/test_gpu_rows_attn.py -x -q -s -p no:cacheprovider > $O/rows_attn_tests.txt 2>&1; echo "rows_attn rc $?" | tee -a $O/summary.txt
grep -E "^\[8 rows|passed|failed|Error|error" $O/rows_attn_tests.txt | tail -20
timeout 300 python -m pytest tests/test_gpu_kernels.py -k "deep_cache" -x -q -p no:cacheprovider > $O/deep_kernel_tests.txt 2>&1; echo "deep kernel tests rc $?" | tee -a $O/summary.txt
tail -3 $O/deep_kernel_tests.txt
timeout 300 python -m pytest tests/test_gpu_reference_anchor.py -k "batched_decode and 8" -x -q -s -p no:cacheprovider > $O/anchor_b8.txt 2>&1; echo "anchor b8 rc $?" | tee -a $O/summary.txt
grep -E "matrix-core decode path|passed|failed" $O/anchor_b8.txt | tail -6
for f in 0 1; do
  timeout 200 python scripts/trace_step.py --batch 8 --lens 300,3858,7300 --options fuse_rows_attn=$f > $O/timeline_b8_fused$f.txt 2>&1; echo "trace fused=$f rc $?" | tee -a $O/summary.txt
done
tail -12 $O/timeline_b8_fused1.txt
timeout 400 python scripts/stress_gemm256.py 600 > $O/stress_gemm256.txt 2>&1; echo "stress rc $?" | tee -a $O/summary.txt
tail -4 $O/stress_gemm256.txt
# ASan build of the library (MA_DEBUG=asan, built in the authoring container): tiny pipeline + the fused-launch tests
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null)
( export MA_DEBUG=asan HSA_XNACK=1 LD_PRELOAD="$ASAN_RT" ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0
  timeout 300 python -m pytest tests/test_gpu_pipeline.py -k "tiny" -x -q -p no:cacheprovider > $O/asan_tiny.txt 2>&1; echo "asan tiny rc $?" | tee -a $O/summary.txt )
tail -5 $O/asan_tiny.txt
wait $CPU_PID; echo "cpu baseline rc $?" | tee -a $O/summary.txt
cat $O/cpu_baseline_full.json | cut -c1-600
