#!/bin/bash
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "bn_folded" -s 2>&1 | grep -v Warning | grep -E "parity|passed|failed|Error|assert" | head -20 > $O/fold_test.txt
for v in 0 1 2 3 4; do
mkdir -p $O/d2
HN_FOLD_DEBUG=$v rocprofv3 --kernel-trace -f csv -d $O/d2/tr -- python tools/prof_train_target.py bf16 64 2 > $O/d2/tr.log 2>&1
python tools/trace_timeline.py $O/d2/tr prep_nhwc4_kernel --list > $O/r5d_fold_dbg$v.txt 2>> $O/d2/tr.log
rm -rf $O/d2/tr
done
