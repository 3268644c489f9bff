#!/bin/bash
# per-dispatch wgrad durations under the three 8-wave tile shapes (B = 64)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 2 3; do
  rm -rf gpurun_out/r24_tl
  HN_WGRAD_W8=$v rocprofv3 --kernel-trace -f csv -d gpurun_out/r24_tl -- python tools/prof_train_target.py bf16 64 2 > gpurun_out/r24.log 2>&1
  python tools/trace_timeline.py gpurun_out/r24_tl prep_nhwc4_kernel --list > gpurun_out/r24_w8_$v.txt 2>> gpurun_out/r24.log
  echo "W8=$v $(sed -n 2p gpurun_out/r24_w8_$v.txt | cut -c1-50)"
done
rm -rf gpurun_out/r24_tl
