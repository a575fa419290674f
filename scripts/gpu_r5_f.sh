#!/bin/bash
# round 5, call F: request placement variants of the fused 8-row launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
timeout 600 python -m pytest tests/test_gpu_rows_attn.py -q -s -p no:cacheprovider -k "bitwise or launch_count" > $O/rows_attn_tests.txt 2>&1; echo "rows_attn rc $?"
grep -E "^\[8 rows, bf16|passed|failed|^E  " $O/rows_attn_tests.txt | cut -c1-600 | tail -8
for early in 2 3 4; do
timeout 200 python scripts/trace_step.py --batch 8 --lens 3858 --options rows_attn_early=$early > $O/timeline_b8_early$early.txt 2>&1
echo "--- rows_attn_early=$early"; grep -v amdgpu.ids $O/timeline_b8_early$early.txt
done
timeout 300 python - > $O/ab_steps.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.checkpoint import synthetic_items
from meshanything_amd.engine import Engine
cfg = MAConfig.full(dtype=DTYPE_BF16, max_batch=8)
eng = Engine(cfg); eng.load_weights(synthetic_items(cfg)); eng.set_option("profile_batch", 8)
for kv in (600, 3858, 7300):
    for rep in range(2):
        row = []
        for early in (0, 1, 2, 3, 4):
            eng.set_option("rows_attn_early", early)
            p = eng.profile_decode(kv, 8)
            row.append(f"early{early}: {1e3 * p['step_ms_graph']:7.1f}")
        print(f"kv {kv:5d} | " + " | ".join(row), flush=True)
PY
grep -v amdgpu.ids $O/ab_steps.txt
