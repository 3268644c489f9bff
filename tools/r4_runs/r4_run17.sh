#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
SWEEP_B=64 SWEEP_ONLY=layer1,layer2.0.conv1 SWEEP_VARIANTS=0,-1 SWEEP_NOASSERT=1 timeout 300 python tools/conv_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r17_sweep_b64.txt
timeout 900 python -m tools.c5_layout 1000 > gpurun_out/r16_c5.json 2> gpurun_out/r16_c5.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r16_c5.json'))
for p in ('f32', 'bf16'):
    print(p, {k: v for k, v in d[p].items() if k != 'per_image'})
PY
