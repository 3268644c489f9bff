#!/bin/bash
python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "b16_vs_reference" -s 2>&1 | grep -E "parity\]|passed|failed|Error|assert" | tail -10 > gpurun_out/t_b16.txt
python -m pytest tests/test_gpu_integration.py -x -q -m gpu -k "eight_rank" -s 2>&1 | grep -E "parity\]|passed|failed|Error|assert|rror" | tail -20 > gpurun_out/t_8rank.txt
