#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_bf16.py -x -q -k "pingpong" 2>&1 | tail -3
STAMP_ONLY=ghc1.0 timeout -k 5 200 python tools/pp_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_6_stamps.txt
W8SHAPES="layer2.0.downsample,layer2.x.conv3,layer3,layer4.0.downsample,layer4.x.conv3,ghc1.0,ghc1.1,ghc2.0,lstm"
SWEEP_NOASSERT=1 SWEEP_ONLY=$W8SHAPES SWEEP_VARIANTS=1,4,5 timeout -k 5 300 python tools/conv_sweep.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-150 | tee gpurun_out/r4_6_sweep.txt
for r in 1 2; do
  for v in 0 1 2; do
    echo "[PP=$v] $(HN_BF16_PP=$v timeout -k 5 200 python bench.py --dtype bf16 --steps 20 --warmup 5 --legs none --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120)" | tee -a gpurun_out/r4_6_ab.txt
  done
done
