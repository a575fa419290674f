#!/bin/bash
# round 4, call M: fold A/B at 8 rows with the new prologues; timeline at 64 rows; full generations at batch 8 / 64
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/prof_step.py --batch 8 --steps 8 --options "use_graph=1;mfma_fold_ln=0;mfma_fold_qkv_max=0;mfma_fold_fc1_max=0" 2>&1 | grep "len"
timeout 300 python scripts/prof_step.py --batch 6 --steps 8 --options "use_graph=1;mfma_fold_ln=0" 2>&1 | grep "len"
echo "== timeline 64 rows"
timeout 300 python scripts/trace_step.py --batch 64 --lens 3858 2>&1 | grep -v amdgpu.ids > gpurun_out/r04m_trace_b64.log; cat gpurun_out/r04m_trace_b64.log
echo "== timeline 16 rows"
timeout 300 python scripts/trace_step.py --batch 16 --lens 3858 2>&1 | grep -v amdgpu.ids > gpurun_out/r04m_trace_b16.log; cat gpurun_out/r04m_trace_b16.log
echo "== full generations"
timeout 600 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04m_b8_800.json 2> gpurun_out/r04m_b8.err; cut -c1-260 gpurun_out/r04m_b8_800.json; echo
timeout 600 python bench.py --batch 64 --sampling --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r04m_cfg3.json 2> gpurun_out/r04m_cfg3.err; cut -c1-260 gpurun_out/r04m_cfg3.json; echo
