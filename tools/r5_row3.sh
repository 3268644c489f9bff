#!/bin/bash
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
timeout 400 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "wgrad or units_locally or golden or b16" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 > $O/row3_test.txt
for m in x 0 x 0; do env $( [ $m = x ] && echo HN_DUMMY=1 || echo HN_WGRAD_ROW3=$m ) timeout 100 python tools/prof_train_target.py bf16 64 6 2>&1 | grep PROF_TRAIN | sed "s/^/row3=$m /"; done >> $O/row3_test.txt
