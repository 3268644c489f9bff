#!/bin/bash
# round-5 diagnostic call 1: per-dispatch timelines (training step B=64, bf16 forward B=32 plain, f32 forward B=32 plain)
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
mkdir -p $O/d1
rocprofv3 --kernel-trace -f csv -d $O/d1/tr -- python tools/prof_train_target.py bf16 64 2 > $O/d1/tr.log 2>&1
python tools/trace_timeline.py $O/d1/tr prep_nhwc4_kernel --list > $O/r5a_train_bf16_B64_timeline_list.txt 2>> $O/d1/tr.log
rocprofv3 --kernel-trace -f csv -d $O/d1/bf -- python tools/prof_target.py bf16 32 3 > $O/d1/bf.log 2>&1
python tools/trace_timeline.py $O/d1/bf stem_pool --list > $O/r5a_bf16_forward_timeline_list.txt 2>> $O/d1/bf.log
rocprofv3 --kernel-trace -f csv -d $O/d1/f32 -- python tools/prof_target.py f32 32 3 > $O/d1/f32.log 2>&1
python tools/trace_timeline.py $O/d1/f32 prep_nhwc4_kernel --list > $O/r5a_f32_forward_timeline_list.txt 2>> $O/d1/f32.log
tail -3 $O/d1/*.log
rm -rf $O/d1/tr $O/d1/bf $O/d1/f32
