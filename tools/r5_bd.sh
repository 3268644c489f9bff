#!/bin/bash
timeout 600 python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err
python - <<'PY'
import json
for l in open('gpurun_out/r5_bench_default.json'):
    if l.startswith('{'):
        d = json.loads(l)
        r = d['roofline']; t = d.get('train_bf16', {})
        print('f32', d['value'], r['frac'], r.get('traffic_stale'), 'bf16', d.get('bf16_mode', {}).get('value'), 'train', t.get('value'), t.get('ms_per_step'), t.get('roofline', {}).get('frac'), t.get('roofline', {}).get('traffic_stale'), 'layout', d.get('layout', {}).get('value'), d.get('leg_seconds'))
PY
