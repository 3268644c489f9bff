#!/bin/bash
O=$(pwd)/gpurun_out
timeout 400 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "pingpong or dw_reuse or forward_bf16" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4 > $O/pp_test.txt
timeout 400 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "bn_fold or golden or b16" -s 2>&1 | grep -E "parity\] fold fwd|parity\] bf16 train step B|passed|failed|Error|assert" | tail -14 >> $O/pp_test.txt
for m in 1 0 1 0; do HN_FOLD_PP=$m timeout 100 python tools/prof_train_target.py bf16 64 6 2>&1 | grep PROF_TRAIN | sed "s/^/foldpp=$m /"; done >> $O/pp_test.txt
