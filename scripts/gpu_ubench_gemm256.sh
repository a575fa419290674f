#!/bin/bash
mkdir -p gpurun_out
timeout 300 ./scripts/ubench_gemm256 2>&1 | tee gpurun_out/ubench_gemm256.txt
