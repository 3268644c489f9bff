#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
SH="ghc1.0,layer3.x.conv1,layer3.x.conv2,layer3.x.conv3,layer2.x.conv3,ghc2.0"
echo "## product"; SWEEP_NOASSERT=1 SWEEP_ONLY=$SH SWEEP_VARIANTS=0,4 timeout -k 5 300 python tools/conv_sweep.py 2>&1 | grep -v "amdgpu.ids\|^#" | cut -c1-120 | tee gpurun_out/r4_7_policy.txt
for P in pol01 pol02 pol03 pol04 pol10 pol11 pol20 pol22; do
  echo "## $P (A policy, B policy: 0 default 1 nt 2 sc1 3 sc0 sc1 4 sc1 nt)" | tee -a gpurun_out/r4_7_policy.txt
  SWEEP_LIB=tools/probe/pp_abl_$P.so SWEEP_NOASSERT=1 SWEEP_ONLY=$SH SWEEP_VARIANTS=0,4 timeout -k 5 300 python tools/conv_sweep.py 2>&1 | grep -v "amdgpu.ids\|^#" | cut -c1-120 | tee -a gpurun_out/r4_7_policy.txt
done
