#!/bin/bash
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "bn_fold or stem_conv_direct or golden" -s 2>&1 | grep -v Warning | grep -E "parity\] fold|parity\] folded|passed|failed|Error|assert" | tail -40 > $O/fold_test.txt
for m in 1 0 1 0; do HN_BN_FOLD=$m python tools/prof_train_target.py bf16 64 6 2>&1 | grep PROF_TRAIN | sed "s/^/fold=$m /"; done >> $O/fold_test.txt
mkdir -p $O/d2
rocprofv3 --kernel-trace -f csv -d $O/d2/tr -- python tools/prof_train_target.py bf16 64 2 > $O/d2/tr.log 2>&1
python tools/trace_timeline.py $O/d2/tr prep_nhwc4_kernel --list > $O/r5f_fold.txt 2>> $O/d2/tr.log
rm -rf $O/d2/tr
