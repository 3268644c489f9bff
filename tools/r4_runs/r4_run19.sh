#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
SWEEP_VARIANTS=0,-1 SWEEP_NOASSERT=1 timeout 600 python tools/conv_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r19_sweep_auto.txt
cat gpurun_out/r19_sweep_auto.txt | cut -c1-120
