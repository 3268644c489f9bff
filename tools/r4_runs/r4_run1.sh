#!/bin/bash
# round-4 GPU call 1: ping-pong conv kernel -- bit-identity tests, per-shape sweep against the round-3 kernels, bf16 forward A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_bf16.py -x -q -k "pingpong or conv_bf16_stage" > gpurun_out/r4_1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_1_pytest.log
tail -5 gpurun_out/r4_1_pytest.log
W8SHAPES="layer2.0.downsample,layer2.x.conv3,layer3,layer4.0.downsample,layer4.x.conv3,ghc1.0,ghc1.1,ghc2.0,ghc2.1,ghc3.0,lstm"
SWEEP_NOASSERT=1 SWEEP_ONLY=$W8SHAPES SWEEP_VARIANTS=0,1,4,5,-1 timeout -k 5 300 python tools/conv_sweep.py > gpurun_out/r4_1_sweep_pp.txt 2>&1
HN_BF16_PP=0 SWEEP_NOASSERT=1 SWEEP_VARIANTS=0,-1 timeout -k 5 300 python tools/conv_sweep.py > gpurun_out/r4_1_sweep_old.txt 2>&1
tail -40 gpurun_out/r4_1_sweep_pp.txt
tail -3 gpurun_out/r4_1_sweep_old.txt
for r in 1 2; do
  for v in 0 1 2; do
    echo "[PP=$v] $(HN_BF16_PP=$v timeout -k 5 200 python bench.py --dtype bf16 --steps 20 --warmup 5 --legs none --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300)" >> gpurun_out/r4_1_ab.txt
  done
done
cat gpurun_out/r4_1_ab.txt
rocprofv3 --kernel-trace -f csv -d gpurun_out/r4_1_tl -- python tools/prof_target.py bf16p 32 6 > gpurun_out/r4_1_tl.log 2>&1
python tools/trace_timeline.py gpurun_out/r4_1_tl stem_pool_bf16_kernel --list > gpurun_out/r4_1_bf16_timeline.txt 2>> gpurun_out/r4_1_tl.log
head -30 gpurun_out/r4_1_bf16_timeline.txt
rm -rf gpurun_out/r4_1_tl
