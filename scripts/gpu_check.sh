#!/bin/bash
# One gpurun call: kernel parity, pipeline parity, smoke.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== device"; python -c "import torch;print(torch.cuda.get_device_name(0), torch.cuda.device_count())"; nproc; free -g | head -2
  echo "== kernels"; timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -60
  echo "== pipeline tiny"; timeout 1500 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --no-header -p no:cacheprovider -k "tiny or weights" 2>&1 | tail -120
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -20
} > gpurun_out/check_a.log 2>&1
tail -c 6000 gpurun_out/check_a.log
