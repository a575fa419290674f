#!/bin/bash
# round 4, call A: the new parity tests (teacher-forced reference anchors on diverse streams, forced tokens), the whole GPU suite, the bench
# line with its new fields, and the rocprofv3 kernel trace of the bench command (looking for the 20 ms oproj_fc1 outlier of round 3).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== new tests"
timeout 900 python -m pytest tests/test_gpu_reference_anchor.py -x -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" | tail -40 | tee gpurun_out/r04a_anchor_tests.txt
echo "== whole suite"
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04a_suite.txt; tail -45 gpurun_out/r04a_suite.txt
echo "== bench"
timeout 900 python bench.py > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
tail -c 4500 gpurun_out/r04a_bench.json; tail -c 600 gpurun_out/r04a_bench.err
echo "== rocprof kernel trace of the bench command"
cd /tmp; rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r4 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batched-table > $R/gpurun_out/r04a_prof.log 2>&1
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/r04a_bench_kernel_stats.csv; done
head -8 $R/gpurun_out/r04a_bench_kernel_stats.csv | cut -c1-220
tail -c 1500 $R/gpurun_out/r04a_prof.log | grep -o '"fused_launch_health[^}]*}[^}]*}[^}]*}'
# the longest dispatches of the fused launches, with their start times: is the 20 ms outlier back, and where in the run does it sit?
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY' | tee $R/gpurun_out/r04a_long_dispatches.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
big = sorted(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)
print("dispatches:", len(rows))
for r in big[:12]:
    print("%10.3f ms at t=%9.3f s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - t0) / 1e9, r["Kernel_Name"][:90]))
fused = [r for r in rows if "oproj_fc1" in r["Kernel_Name"] or "qkv_attn" in r["Kernel_Name"]]
long_ = [r for r in fused if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
print("fused launches:", len(fused), "of them longer than 100 us:", len(long_))
for r in long_[:20]:
    print("   %10.3f ms at t=%9.3f s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - t0) / 1e9, r["Kernel_Name"][:60]))
PY
