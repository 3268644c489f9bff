#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/pipe_bench.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r26_pipe_bench.txt
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "replicated" 2>&1 | tail -3
