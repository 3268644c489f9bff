#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -25 | tee gpurun_out/r4_9_pytest.log
for r in 1 2; do
  for v in 0 1; do
    echo "[SPLITK=$v] $(HN_BF16_SPLITK=$v timeout -k 5 200 python bench.py --dtype bf16 --steps 20 --warmup 5 --legs none --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120)" | tee -a gpurun_out/r4_9_ab.txt
  done
done
rocprofv3 --kernel-trace -f csv -d gpurun_out/r4_9_tl -- python tools/prof_target.py bf16p 32 6 > gpurun_out/r4_9_tl.log 2>&1
python tools/trace_timeline.py gpurun_out/r4_9_tl stem_pool_bf16_kernel --list > gpurun_out/r4_9_bf16_timeline.txt 2>> gpurun_out/r4_9_tl.log
head -32 gpurun_out/r4_9_bf16_timeline.txt
rm -rf gpurun_out/r4_9_tl
