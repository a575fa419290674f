#!/bin/bash
# round 4, call G: whole suite on the tree with the 256 x 256 GEMM in the pipeline, bench line, bench at batch 8 / config 3 / config 5
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== whole suite"
timeout 2000 python -m pytest tests/ -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04g_suite.txt
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r04g_suite.txt | tail -12
grep -E "^E  " gpurun_out/r04g_suite.txt | head -20
echo "== bench"
timeout 900 python bench.py > gpurun_out/r04g_bench.json 2> gpurun_out/r04g_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04g_bench.json"))
for k in ("value", "ms_per_step", "phases_ms", "dense_phases", "encoder_max_abs_err", "tokens_distinct", "fused_launch_health", "measured_peaks", "fp32_exact", "fp16_policy", "batched_decode_steps"):
    print(k, "=", json.dumps(d.get(k)))
print("roofline frac", d["roofline"]["frac"], "step ms", d["roofline"]["decode_step_ms_graph"], {k: v["avg_launch_us"] for k, v in d["roofline"]["classes"].items()})
PY
tail -c 300 gpurun_out/r04g_bench.err
echo "== configs"
timeout 600 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04g_b8_800.json 2> gpurun_out/r04g_b8.err; cut -c1-300 gpurun_out/r04g_b8_800.json
timeout 600 python bench.py --batch 64 --sampling --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r04g_cfg3_b64_sampling.json 2> gpurun_out/r04g_cfg3.err; cut -c1-300 gpurun_out/r04g_cfg3_b64_sampling.json
timeout 600 python bench.py --batch 8 --faces 1600 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r04g_cfg5_b8_1600.json 2> gpurun_out/r04g_cfg5.err; cut -c1-300 gpurun_out/r04g_cfg5_b8_1600.json
