#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/trace_step.py ${TRACE_ARGS:-} > gpurun_out/trace_step.log 2>&1
grep -v amdgpu.ids gpurun_out/trace_step.log | tail -40
