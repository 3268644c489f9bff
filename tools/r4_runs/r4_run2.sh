#!/bin/bash
# round-4 GPU call 2: k-half ping-pong schedule -- bit-identity, sweep vs the 8-wave two-stage kernel, loop ablations, forward A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_bf16.py -x -q -k "pingpong or conv_bf16_stage" > gpurun_out/r4_2_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_2_pytest.log
tail -5 gpurun_out/r4_2_pytest.log
W8SHAPES="layer2.0.downsample,layer2.x.conv3,layer3,layer4.0.downsample,layer4.x.conv3,ghc1.0,ghc1.1,ghc2.0,lstm"
SWEEP_NOASSERT=1 SWEEP_ONLY=$W8SHAPES SWEEP_VARIANTS=0,1,4,5 timeout -k 5 300 python tools/conv_sweep.py > gpurun_out/r4_2_sweep_pp.txt 2>&1
cat gpurun_out/r4_2_sweep_pp.txt | cut -c1-150
tools/pp_ablate.sh run "0 1 2 4 8 3 5 6 7 12 16 ord" "ghc1.0,layer3.x.conv1,layer3.x.conv2,layer3.x.conv3" > gpurun_out/r4_2_ablate.txt 2>&1
cat gpurun_out/r4_2_ablate.txt | cut -c1-120
for r in 1 2; do
  for v in 0 1; do
    echo "[PP=$v] $(HN_BF16_PP=$v timeout -k 5 200 python bench.py --dtype bf16 --steps 20 --warmup 5 --legs none --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120)" >> gpurun_out/r4_2_ab.txt
  done
done
cat gpurun_out/r4_2_ab.txt
