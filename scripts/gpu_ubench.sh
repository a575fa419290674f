#!/bin/bash
# GPU box: build one microbenchmark of scripts/ (hipcc is on the box) and run it under a timeout -> gpurun_out/<tag>.log
# usage: scripts/gpu_ubench.sh <name without .hip> [tag] [timeout seconds] [args...]
name=$1; tag=${2:-$1}; tmo=${3:-300}; shift; shift; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
{
hipcc --offload-arch=gfx950 -O3 -DNDEBUG -std=c++17 -Wno-unused-result scripts/$name.hip -o /tmp/$name 2>&1 | grep -v warning | tail -5
timeout $tmo /tmp/$name "$@" 2>&1
echo "rc=$?"
} > gpurun_out/$tag.log 2>&1
tail -c 6000 gpurun_out/$tag.log
