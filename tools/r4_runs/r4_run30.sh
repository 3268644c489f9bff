#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d gpurun_out/r30_tl -- python tools/prof_target.py bf16p 32 6 > gpurun_out/r30.log 2>&1
python tools/trace_timeline.py gpurun_out/r30_tl stem_pool_bf16_kernel --list > gpurun_out/r4_bf16_pipelined_timeline.txt 2>> gpurun_out/r30.log
head -3 gpurun_out/r4_bf16_pipelined_timeline.txt
rm -rf gpurun_out/r30_tl
