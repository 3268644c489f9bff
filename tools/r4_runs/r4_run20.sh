#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  env "$@" timeout 300 python bench.py --dtype bf16 --legs none --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', d['value'], d['ms_per_step'])"
}
for i in 1 2; do
  run HN_X=0
  run HN_BF16_SPLITK=2
  run HN_BF16_MIN_TILES=128
  run HN_BF16_MIN_TILES=128 HN_BF16_SPLITK=2
  run HN_BF16_MIN_TILES=64 HN_BF16_SPLITK=2
done
