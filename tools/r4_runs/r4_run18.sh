#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do
  for v in normal low high; do
    HN_BRANCH_PRIORITY=$v timeout 300 python bench.py --dtype bf16 --legs none --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('branch priority $v', d['value'], d['ms_per_step'])"
  done
done
