#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py -q -m gpu 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -8 | cut -c1-400
rocprofv3 --kernel-trace -f csv -d gpurun_out/r35_tl -- python tools/prof_train_target.py bf16 64 2 > gpurun_out/r35.log 2>&1
python tools/trace_timeline.py gpurun_out/r35_tl prep_nhwc4_kernel > gpurun_out/r35_t.txt 2>> gpurun_out/r35.log
head -3 gpurun_out/r35_t.txt | cut -c1-140; grep "upsample_flatten_bwd\|maxpool\|affine_act_bn_pool" gpurun_out/r35_t.txt | cut -c1-120
rm -rf gpurun_out/r35_tl
