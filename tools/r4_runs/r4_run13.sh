#!/bin/bash
# round 4, call 13: A/B of the 64-column dw-reuse kernel inside the whole forward + kernel stats
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
  for v in 1 0; do
    HN_BF16_DWR64=$v timeout 300 python bench.py --dtype bf16 --legs none --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('DWR64=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_r13/stats -- python bench.py --dtype bf16 --steps 5 --warmup 2 --legs none --no-cpu-baseline > /dev/null 2> gpurun_out/r13_stats.log
python tools/prof_summary.py stats gpurun_out/prof_r13/stats "rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --steps 5 --warmup 2 --legs none" > gpurun_out/r13_bf16_kernel_stats.txt 2>> gpurun_out/r13_stats.log
head -50 gpurun_out/r13_bf16_kernel_stats.txt
rm -rf gpurun_out/prof_r13
