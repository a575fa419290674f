#!/bin/bash
# round 5, call H: L2 prefetch of the next layer's first operands by the idle blocks of the MLP launch: A/B + bitwise + distributions
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
timeout 300 python - > $O/ab_prefetch.txt 2>&1 <<'PY'
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
        for pf, early in ((0, 3), (8, 3), (9, 3), (1, 3), (8, 4), (9, 4)):
            eng.set_option("rows_mlp_prefetch", pf); eng.set_option("rows_attn_early", early)
            p = eng.profile_decode(kv, 8)
            row.append(f"pf{pf} early{early}: {1e3 * p['step_ms_graph']:7.1f}")
        print(f"kv {kv:5d} | " + " | ".join(row), flush=True)
print("timeouts", eng.get_option("xchg_timeouts"))
PY
grep -v amdgpu.ids $O/ab_prefetch.txt

timeout 200 python scripts/trace_step.py --batch 8 --lens 3858 --options rows_mlp_prefetch=8 > $O/timeline_b8_prefetch8.txt 2>&1; grep -v amdgpu.ids $O/timeline_b8_prefetch8.txt
timeout 200 python scripts/trace_step.py --batch 8 --lens 3858 --options rows_mlp_prefetch=9 > $O/timeline_b8_prefetch9.txt 2>&1; grep -v amdgpu.ids $O/timeline_b8_prefetch9.txt
