#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py -q -m gpu 2>&1 | grep -E "^E |passed|failed|^FAILED" | head -20 | cut -c1-700
