#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_bf16.py -x -q -k "pingpong" > gpurun_out/r4_4_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_4_pytest.log
tail -5 gpurun_out/r4_4_pytest.log
W8SHAPES="layer2.0.downsample,layer2.x.conv3,layer3,layer4.0.downsample,layer4.x.conv3,ghc1.0,ghc1.1,ghc2.0,lstm"
SWEEP_NOASSERT=1 SWEEP_ONLY=$W8SHAPES SWEEP_VARIANTS=0,1,4,5 timeout -k 5 300 python tools/conv_sweep.py 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee gpurun_out/r4_4_sweep_pp.txt
STAMP_ONLY=ghc1.0,layer3.x.conv2 timeout -k 5 200 python tools/pp_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_4_stamps.txt
STAMP_LO=3 STAMP_HI=12 STAMP_ONLY=layer3.x.conv3,layer3.x.conv1,layer2.x.conv3 timeout -k 5 200 python tools/pp_stamps.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_4_stamps.txt
tools/pp_ablate.sh run "1 2 3 8" "ghc1.0,layer3.x.conv1,layer3.x.conv2,layer3.x.conv3" 2>&1 | cut -c1-120 | tee gpurun_out/r4_4_ablate.txt
for r in 1 2; do
  for v in 0 1; do
    echo "[PP=$v] $(HN_BF16_PP=$v timeout -k 5 200 python bench.py --dtype bf16 --steps 20 --warmup 5 --legs none --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120)" | tee -a gpurun_out/r4_4_ab.txt
  done
done
