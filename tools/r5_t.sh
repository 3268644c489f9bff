#!/bin/bash
python -m pytest tests/test_gpu_train.py tests/test_gpu_dataset.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -10 > gpurun_out/t_all_train.txt
for m in 1 1; do python tools/prof_train_target.py bf16 64 6 2>&1 | grep PROF_TRAIN; done >> gpurun_out/t_all_train.txt
