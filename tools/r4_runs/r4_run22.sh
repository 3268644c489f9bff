#!/bin/bash
# wgrad launch-parameter sweep at B = 64 (r2 tuned them at B = 32): family totals per step from the kernel trace
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  rm -rf gpurun_out/r22_tl
  env "$@" rocprofv3 --kernel-trace -f csv -d gpurun_out/r22_tl -- python tools/prof_train_target.py bf16 64 2 > gpurun_out/r22.log 2>&1
  python tools/trace_timeline.py gpurun_out/r22_tl prep_nhwc4_kernel > gpurun_out/r22_t.txt 2>> gpurun_out/r22.log
  echo "== $*: $(head -2 gpurun_out/r22_t.txt | tail -1 | cut -c1-60)"
  grep "conv_wgrad" gpurun_out/r22_t.txt | awk '{t+=$(NF-1); printf "   %s %s %s\n", $1" "$2" "$3" "$4" "$5, $(NF-2), $(NF-1)} END {print "   wgrad total ms", t}'
}
run HN_X=0
run HN_WGRAD_W8_WGS=512
run HN_WGRAD_H_WGS=1024
run HN_WGRAD_W8=4
run HN_WGRAD_W8=4 HN_WGRAD_W8_WGS=512
rm -rf gpurun_out/r22_tl
