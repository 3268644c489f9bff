#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -s -k "pooled_gradient or fused_stem" 2>&1 | grep -E "^E  |passed|failed|^FAILED|pooled gradient" | head -12 | cut -c1-900
