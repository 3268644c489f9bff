#!/bin/bash
# round 4, call 12: the 64-column dw-reuse kernel (parity, per-shape sweep, bench) + the full GPU suite after the boundary changes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "dw_reuse or pingpong or split_k" > gpurun_out/r12_dwr_tests.log 2>&1
tail -3 gpurun_out/r12_dwr_tests.log
SWEEP_ONLY=layer1.x.conv2,layer1.0.conv2,ghc0.2,ghc0.3,ghc1.3 SWEEP_VARIANTS=0,8,-1 SWEEP_NOASSERT=1 timeout 300 python tools/conv_sweep.py > gpurun_out/r12_sweep64.txt 2>&1
cat gpurun_out/r12_sweep64.txt
timeout 600 python bench.py > gpurun_out/r12_bench.json 2> gpurun_out/r12_bench.err
python - <<'PY'
import json
for l in open('gpurun_out/r12_bench.json'):
    if l.startswith('{'):
        d = json.loads(l)
        print('bf16', d['value'], d['ms_per_step'], d['roofline']['frac'], '| f32', d.get('fp32_forward', {}).get('value'), '| train', d.get('train_bf16', {}).get('value'), '| layout', d.get('layout', {}).get('value'))
PY
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r12_gpu_suite.log 2>&1
tail -5 gpurun_out/r12_gpu_suite.log
