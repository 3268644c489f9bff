#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
run() {
  env "$@" timeout 300 python bench.py --dtype bf16 --legs none --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', d['value'], d['ms_per_step'])"
}
for i in 1 2; do
  run HN_D64_GRID=256
  run HN_D64_GRID=224
  run HN_D64_GRID=208
  run HN_D64_GRID=240
done
