#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/r4_10_pytest.log
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_10_bench.json 2> gpurun_out/r4_10_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_10_bench.json').read().strip().splitlines()[-1])
print('f32', d['value'], d['roofline']['frac'])
b=d.get('bf16_mode',{})
print('bf16', b.get('value'), b.get('roofline'), 'plain', b.get('plain_forward'))
print('train', d.get('train_bf16',{}).get('value'), d.get('train_bf16',{}).get('ms_per_step'))
print('layout', d.get('layout',{}).get('value'))
print('legs', d.get('leg_seconds'))
PY
