#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/poison_hunt.py 1 1 > gpurun_out/r6_hunt2_1_1.txt 2>&1; echo "hunt rc $?"; grep "\[hunt\]" gpurun_out/r6_hunt2_1_1.txt | tail -7
timeout 600 python tools/poison_hunt.py 2 1 > gpurun_out/r6_hunt2_2_1.txt 2>&1; echo "hunt rc $?"; grep "\[hunt\]" gpurun_out/r6_hunt2_2_1.txt | tail -7
timeout 600 python tools/soak_determinism.py 150 1 > gpurun_out/r6_soak2.txt 2>&1; echo "soak rc $?"; grep "\[soak\]" gpurun_out/r6_soak2.txt | tail -5
timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -s > gpurun_out/r6_train_tests.txt 2>&1; echo "train tests rc $?"; grep -n "passed\|failed" gpurun_out/r6_train_tests.txt | tail -2; grep -n "trained ckpt" gpurun_out/r6_train_tests.txt
for v in "HN_FOLD_SLAB=0" "HN_FOLD_SLAB=1" "HN_FOLD_SLAB=0" "HN_FOLD_SLAB=1"; do
  echo "# $v"; env $v timeout 400 python bench.py --legs train --no-cpu-baseline --steps 3 --train-steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train_bf16']; print(t['value'], t['ms_per_step'])"
done
