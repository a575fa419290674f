#!/bin/bash
# round 4, call B: exact encoder under the bf16 policy (cfg.enc_exact), the fp32 matrix-core attention, diagnostics of the fused launches
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== anchor + kernel tests"
timeout 900 python -m pytest tests/test_gpu_reference_anchor.py tests/test_gpu_kernels.py -q -m gpu -s -k "anchor or reference or attention or forced" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tee gpurun_out/r04b_anchor_tests.txt | tail -40
echo "== whole suite"
timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04b_suite.txt; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r04b_suite.txt | tail -30
echo "== bench"
timeout 900 python bench.py > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04b_bench.json"))
for k in ("value", "ms_per_step", "phases_ms", "dense_phases", "encoder_max_abs_err", "tokens_distinct", "fused_launch_health", "measured_peaks", "fp32_exact", "batched_decode_steps"):
    print(k, "=", json.dumps(d.get(k)))
print("roofline frac", d["roofline"]["frac"], "step ms", d["roofline"]["decode_step_ms_graph"])
PY
tail -c 400 gpurun_out/r04b_bench.err
