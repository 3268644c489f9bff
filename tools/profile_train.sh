#!/bin/bash
# rocprofv3 evidence for the bf16 training step at B = 64 (run on the GPU box): kernel stats + dispatch timeline of ONE step, and separate
# counter passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) over tools/prof_train_target.py.  Usage: tools/profile_train.sh TAG
set -u
TAG=$1
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/prof_${TAG}_train
mkdir -p $OUT
N=3
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- python tools/prof_train_target.py bf16 64 $N > $OUT/stats.log 2>&1
python tools/prof_summary.py stats $OUT/stats "rocprofv3 --kernel-trace --stats -- python tools/prof_train_target.py bf16 64 $N ($((N + 1)) steps)" > gpurun_out/${TAG}_train_bf16_B64_kernel_stats.txt 2>> $OUT/stats.log
python tools/trace_timeline.py $OUT/stats prep_nhwc4_kernel > gpurun_out/${TAG}_train_bf16_B64_timeline.txt 2>> $OUT/stats.log
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  K=$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pmc_$K -- python tools/prof_train_target.py bf16 64 $N > $OUT/pmc_$K.log 2>&1
  python tools/prof_summary.py counters $OUT/pmc_$K $((N + 1)) "rocprofv3 --kernel-trace --pmc $C -- python tools/prof_train_target.py bf16 64 $N" > gpurun_out/${TAG}_train_pmc_$K.json 2>> $OUT/pmc_$K.log
done
python tools/merge_pmc_train.py gpurun_out $TAG > gpurun_out/${TAG}_pmc_train.json
ls -la gpurun_out/${TAG}_train_* gpurun_out/${TAG}_pmc_train.json
rm -rf $OUT
