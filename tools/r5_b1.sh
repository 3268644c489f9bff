#!/bin/bash
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
mkdir -p $O/d3
for p in f32 bf16; do
m=prep_nhwc4_kernel; [ $p = bf16 ] && m=stem_pool
rocprofv3 --kernel-trace -f csv -d $O/d3/$p -- python tools/prof_target.py $p 1 6 > $O/d3/$p.log 2>&1
python tools/trace_timeline.py $O/d3/$p $m --list > $O/r5_b1_${p}_timeline.txt 2>> $O/d3/$p.log
rm -rf $O/d3/$p
done
