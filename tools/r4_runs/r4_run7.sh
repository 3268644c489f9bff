#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_bf16.py -q -k "dw_reuse or pingpong" 2>&1 | tail -25 | tee gpurun_out/r4_8_pytest.log
SWEEP_NOASSERT=1 SWEEP_ONLY="conv2,ghc0.0,ghc0.1,ghc1.0,ghc1.1,ghc2.0,ghc2.1,ghc3.0,ghc3.1" SWEEP_VARIANTS=0,4,6,7,-1 timeout -k 5 300 python tools/conv_sweep.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-170 | tee gpurun_out/r4_8_sweep.txt
for r in 1 2; do
  for v in 0 1; do
    echo "[DWR=$v] $(HN_BF16_DWR=$v timeout -k 5 200 python bench.py --dtype bf16 --steps 20 --warmup 5 --legs none --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120)" | tee -a gpurun_out/r4_8_ab.txt
  done
done
