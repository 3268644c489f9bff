#!/bin/bash
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
mkdir -p $O/d3
for p in f32; do
timeout 120 rocprofv3 --kernel-trace -f csv -d $O/d3/$p -- python tools/prof_target.py $p 1 6 > $O/d3/$p.log 2>&1
timeout 60 python tools/trace_timeline.py $O/d3/$p prep_nhwc4_kernel --list > $O/r5_b1_${p}_timeline.txt 2>> $O/d3/$p.log
rm -rf $O/d3/$p
done
