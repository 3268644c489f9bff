#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --mode train --dtype bf16 --batch 64 --steps 6 --warmup 2 > gpurun_out/r21_train.json 2> gpurun_out/r21_train.err
python - <<'PY'
import json
for l in open('gpurun_out/r21_train.json'):
    if l.startswith('{'):
        d = json.loads(l)
        print('train', d['value'], d['ms_per_step'], d['roofline']['frac'], d['final_loss'], d.get('loss_curve_check'))
PY
rocprofv3 --kernel-trace -f csv -d gpurun_out/r21_tl -- python tools/prof_train_target.py bf16 64 3 > gpurun_out/r21_train.log 2>&1
python tools/trace_timeline.py gpurun_out/r21_tl prep_nhwc4_kernel --list > gpurun_out/r21_train_bf16_B64_timeline.txt 2>> gpurun_out/r21_train.log
head -12 gpurun_out/r21_train_bf16_B64_timeline.txt | cut -c1-150
rm -rf gpurun_out/r21_tl
