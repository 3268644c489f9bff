#!/bin/bash
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
mkdir -p $O/d5
timeout 150 rocprofv3 --kernel-trace -f csv -d $O/d5/tr -- python tools/prof_train_target.py bf16 64 2 > $O/d5/tr.log 2>&1
timeout 60 python tools/trace_timeline.py $O/d5/tr prep_nhwc4_kernel --list > $O/r5h_train.txt 2>> $O/d5/tr.log
rm -rf $O/d5/tr
