#!/bin/bash
O=$(pwd)/gpurun_out
timeout 500 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "bn_fold or golden or b16 or loss_curve or segmented" -s 2>&1 | grep -E "parity\] folded|parity\] bf16 train step B|passed|failed|Error|assert" | tail -8 > $O/red_test.txt
for m in 1 0 1 0; do HN_FOLD_REDUCE=$m timeout 100 python tools/prof_train_target.py bf16 64 6 2>&1 | grep PROF_TRAIN | sed "s/^/red=$m /"; done >> $O/red_test.txt
