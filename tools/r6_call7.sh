#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "stem or forward or golden or batch" -s > gpurun_out/r6_stem_tests.txt 2>&1; echo "tests rc $?"; grep -n "fused f32 stem\|passed\|failed\|Error" gpurun_out/r6_stem_tests.txt | head -30
for v in "HN_F32_STEM_POOL=0" "HN_F32_STEM_POOL=1" "HN_F32_STEM_POOL=0" "HN_F32_STEM_POOL=1"; do
  echo "# $v"; env $v timeout 300 python bench.py --legs none --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('box'), d['roofline'].get('frac_of_box_measured_mfma'), d['roofline']['breakdown']['stem(prep+conv_igemm+maxpool)'])"
done
