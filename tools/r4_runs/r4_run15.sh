#!/bin/bash
# round 4, call 15: scale/shift kept in registers (PP / DWR): parity + forward A/B vs HEAD~; fresh training-step timeline
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "dw_reuse or pingpong or split_k" 2>&1 | tail -2
for i in 1 2; do
  timeout 300 python bench.py --dtype bf16 --legs none --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bf16 forward', d['value'], d['ms_per_step'])"
done
rocprofv3 --kernel-trace -f csv -d gpurun_out/r15_tl -- python tools/prof_train_target.py bf16 64 3 > gpurun_out/r15_train.log 2>&1
python tools/trace_timeline.py gpurun_out/r15_tl prep_nhwc4_kernel --list > gpurun_out/r15_train_bf16_B64_timeline.txt 2>> gpurun_out/r15_train.log
head -40 gpurun_out/r15_train_bf16_B64_timeline.txt | cut -c1-150
rm -rf gpurun_out/r15_tl
