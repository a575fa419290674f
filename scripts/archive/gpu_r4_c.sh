#!/bin/bash
# round 4, call C: the fp16 policy (MA_DTYPE_F16) -- kernel tests in both 16-bit formats, pipeline + reference anchors, bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== whole suite"
timeout 2000 python -m pytest tests/ -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04c_suite.txt
grep -E "^\[fp16|^\[bf16/|^\[fp32/|passed|failed|^FAILED|^ERROR" gpurun_out/r04c_suite.txt | tail -40
grep -E "^E  " gpurun_out/r04c_suite.txt | head -30
echo "== bench"
timeout 900 python bench.py > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c_bench.json"))
for k in ("value", "ms_per_step", "phases_ms", "encoder_max_abs_err", "tokens_distinct", "fp32_exact", "fp16_policy"):
    print(k, "=", json.dumps(d.get(k)))
PY
tail -c 300 gpurun_out/r04c_bench.err
timeout 600 python bench.py --dtype fp16 --no-cpu-baseline --no-batched-table > gpurun_out/r04c_bench_fp16.json 2> gpurun_out/r04c_bench_fp16.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c_bench_fp16.json"))
print("fp16 bench: value", d["value"], "roofline frac", d["roofline"]["frac"], "step", d["roofline"]["decode_step_ms_graph"], "enc err", d["encoder_max_abs_err"], "distinct", d["tokens_distinct"])
PY
