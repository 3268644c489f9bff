#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_integration.py -q -x -k "async or pipelined or stream or inference" 2>&1 | tail -4 | tee gpurun_out/r4_11_pytest.log
timeout -k 5 300 python tools/ab_option.py defer_join 30 2>&1 | grep -v amdgpu | tee gpurun_out/r4_11_ab_defer.txt
