#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "replicated" 2>&1 | grep -v "^$" | tail -30 | cut -c1-400
