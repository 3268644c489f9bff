#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --force-rccl --no-cpu-baseline > gpurun_out/r6_force_rccl.json 2> gpurun_out/r6_force_rccl.err; echo "force-rccl rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6_force_rccl.json') if l.startswith('{')][-1])
print('force-rccl: f32', d['value'], d['roofline']['frac'], 'bf16', d.get('bf16_mode',{}).get('value'), 'train', d.get('train_bf16',{}).get('value'), 'layout', d.get('layout',{}).get('value'), 'box', d.get('box'))
PY
HN_POISON_WS=1 timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r6_gpu_tests_poisoned.txt 2>&1; echo "suite(engine poison) rc $?"; grep -n "passed\|failed" gpurun_out/r6_gpu_tests_poisoned.txt | tail -2
HN_POISON_WS=7f timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r6_gpu_tests_poisoned_7f.txt 2>&1; echo "suite(engine poison 7f) rc $?"; grep -n "passed\|failed" gpurun_out/r6_gpu_tests_poisoned_7f.txt | tail -2
