#!/bin/bash
python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | grep -v Warning | tail -8 > gpurun_out/train_tests.txt
