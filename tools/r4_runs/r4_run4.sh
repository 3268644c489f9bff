#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_bf16.py -x -q -k "pingpong" 2>&1 | tail -3
STAMP_ONLY=ghc1.0,layer3.x.conv1 timeout -k 5 200 python tools/pp_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_5_stamps.txt
SH="ghc1.0,layer3.x.conv1,layer3.x.conv2,layer3.x.conv3,layer2.x.conv3"
echo "## product"; SWEEP_NOASSERT=1 SWEEP_ONLY=$SH SWEEP_VARIANTS=1,4,5 timeout -k 5 300 python tools/conv_sweep.py 2>&1 | grep -v "amdgpu.ids\|^#" | cut -c1-150 | tee gpurun_out/r4_5_sweep.txt
echo "## lgk before every barrier"; SWEEP_LIB=tools/probe/pp_abl_lgkb.so SWEEP_NOASSERT=1 SWEEP_ONLY=$SH SWEEP_VARIANTS=4,5 timeout -k 5 300 python tools/conv_sweep.py 2>&1 | grep -v "amdgpu.ids\|^#" | cut -c1-150 | tee -a gpurun_out/r4_5_sweep.txt
