#!/bin/bash
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
timeout 400 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "bn_fold or golden or b16 or loss_curve" -s 2>&1 | grep -E "parity\] folded|parity\] bf16 train step B|passed|failed|Error|assert" | tail -8 > $O/fa_test.txt
for m in 1 0 1 0; do HN_FOLD_FUSEA=$m timeout 100 python tools/prof_train_target.py bf16 64 6 2>&1 | grep PROF_TRAIN | sed "s/^/fusea=$m /"; done >> $O/fa_test.txt
