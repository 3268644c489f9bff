#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py -q -m gpu -s 2>&1 | grep -E "^E  |passed|failed|^FAILED|dual BatchNorm" | head -8 | cut -c1-400
for v in 0 1; do
HN_FUSE_BN_DUAL=$v timeout 600 python bench.py --mode train --dtype bf16 --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('dual=$v train', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('loss_curve_check',{}).get('max_rel_diff'))"
done
