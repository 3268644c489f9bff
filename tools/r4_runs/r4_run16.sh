#!/bin/bash
# round 4, call 16: device labels (parity + suite), 2-rank bench with bf16 wire, config-5 seam-aware bf16 figures
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dataset.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -8
timeout 900 python -m pytest tests/test_gpu_integration.py -x -q -m gpu -k "two_rank_train or traincurve or loss_curve or train_py" 2>&1 | tail -5
timeout 900 python -m tools.c5_layout 1000 > gpurun_out/r16_c5.json 2> gpurun_out/r16_c5.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r16_c5.json'))
for p in ('f32', 'bf16'):
    print(p, {k: v for k, v in d[p].items() if k != 'per_image'})
PY
timeout 600 python bench.py --mode train --dtype bf16 --batch 64 --steps 6 --warmup 2 > gpurun_out/r16_train.json 2> gpurun_out/r16_train.err
python - <<'PY'
import json
for l in open('gpurun_out/r16_train.json'):
    if l.startswith('{'):
        d = json.loads(l)
        print('train', d['value'], d['ms_per_step'], d['host_data_pipeline_ms_per_step'], d['host_half_ms_per_batch'], d['config']['labels'])
PY
