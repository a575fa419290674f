#!/bin/bash
# Round-5 evidence, last pass (rows_attn_early = 6): the GPU suite, smoke, the MA_EXPERIMENTAL tests, bench.py (defaults), bench.py --batch 8 with its
# kernel trace, config 5, the 8-row timeline.
mkdir -p gpurun_out/r5final3; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5final3
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids > $O/suite.txt; tail -4 $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
MA_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_persist.py tests/test_gpu_rows_fused.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids > $O/suite_experimental.txt; tail -2 $O/suite_experimental.txt
echo "== bench (defaults)"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "phases_ms", "batched_decode_steps", "fused_launch_health"):
    print(k, "=", json.dumps(d.get(k)))
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if k != "classes"}))
PY
echo "== batch 8"
timeout 600 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > $O/b8_800.json 2> $O/b8.err; cut -c1-300 $O/b8_800.json; echo
timeout 600 python bench.py --batch 8 --faces 1600 --steps 1 --warmup 0 --no-cpu-baseline > $O/cfg5_b8_1600.json 2> $O/cfg5.err; cut -c1-300 $O/cfg5_b8_1600.json; echo
cd /tmp; rm -rf /tmp/prof8
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof8 -o r5 --output-format csv -- python $R/bench.py --batch 8 --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_b8.json 2> $O/prof_b8.log
for f in $(find /tmp/prof8 -name "*kernel_stats*.csv"); do cp $f $O/bench_batch8_kernel_stats.csv; done
head -5 $O/bench_batch8_kernel_stats.csv | cut -c1-200
cd $R
timeout 300 python scripts/trace_step.py --batch 8 --lens 300,3858,7300 --dist 2>&1 | grep -v amdgpu.ids > $O/trace_b8.txt; grep -E "len|attn |fc1 " $O/trace_b8.txt
