#!/bin/bash
O=$(pwd)/gpurun_out
timeout 400 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "pingpong or dw_reuse or forward_bf16" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4 > $O/pp_test.txt
timeout 400 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "bn_fold or golden or b16" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4 >> $O/pp_test.txt
