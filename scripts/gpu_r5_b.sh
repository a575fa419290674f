#!/bin/bash
# round 5, call B: which fused configuration disagrees with which (diag), the all-gather microbenchmark, A/B of the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }
timeout 300 python scripts/diag_rows_attn.py 96 > $O/diag.txt 2>&1; echo "diag rc $?"
grep -v amdgpu.ids $O/diag.txt | tail -30
timeout 120 scripts/ubench_allgather > $O/ubench_allgather.txt 2>&1; echo "ubench rc $?"
cat $O/ubench_allgather.txt
timeout 300 python - > $O/ab_steps.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from meshanything_amd.config import MAConfig, DTYPE_BF16
from meshanything_amd.checkpoint import synthetic_items
from meshanything_amd.engine import Engine
cfg = MAConfig.full(dtype=DTYPE_BF16, max_batch=8)
eng = Engine(cfg); eng.load_weights(synthetic_items(cfg)); eng.set_option("profile_batch", 8)
for kv in (600, 3858, 7300):
    for rep in range(2):
        row = []
        for attn, mlp, early in ((0, 0, 1), (1, 0, 1), (1, 0, 0), (0, 1, 1), (1, 1, 1), (1, 1, 0)):
            eng.set_option("fuse_rows_attn", attn); eng.set_option("fuse_rows_mlp", mlp); eng.set_option("rows_attn_early", early)
            p = eng.profile_decode(kv, 8)
            row.append(f"attn{attn} mlp{mlp} early{early}: {1e3 * p['step_ms_graph']:7.1f} us ({sum(p['launches'].values()) // 8} launches)")
        print(f"kv {kv:5d} | " + " | ".join(row), flush=True)
print("timeouts", eng.get_option("xchg_timeouts"), "fallbacks", eng.get_option("chain_fallbacks"))
PY
echo "ab rc $?"; grep -v amdgpu.ids $O/ab_steps.txt
timeout 200 python scripts/trace_step.py --batch 8 --lens 3858 --options fuse_rows_attn=1,fuse_rows_mlp=1,rows_attn_early=0 > $O/timeline_b8_both_late.txt 2>&1
timeout 200 python scripts/trace_step.py --batch 8 --lens 3858 --options fuse_rows_attn=1,fuse_rows_mlp=1,rows_attn_early=1 > $O/timeline_b8_both_early.txt 2>&1
grep -v amdgpu.ids $O/timeline_b8_both_late.txt; grep -v amdgpu.ids $O/timeline_b8_both_early.txt
