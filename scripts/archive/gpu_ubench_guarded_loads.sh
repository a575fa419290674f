#!/bin/bash
mkdir -p gpurun_out
timeout 120 ./scripts/ubench_guarded_loads 2>&1 | tee gpurun_out/r04_ubench_guarded_loads.txt
