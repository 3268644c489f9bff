#!/bin/bash
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "bn_folded or dual_batchnorm or matches_reference_golden or loss_curve" 2>&1 | tail -15
for m in 1 0 1 0; do HN_BN_FOLD=$m python tools/prof_train_target.py bf16 64 6 2>&1 | grep PROF_TRAIN | sed "s/^/fold=$m /"; done
