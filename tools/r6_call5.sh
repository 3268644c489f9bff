#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_streams3.txt; : > $O
for v in "HN_X=0" "HN_HEAD_PRIO=0" "HN_STREAMS=cumask" "HN_STREAM_SKIP=1" "HN_STREAM_SKIP=2" "HN_STREAM_SKIP=3" "HN_STREAM_SKIP=4" "HN_STREAM_SKIP=5"; do
  echo "# env $v" >> $O
  env $v timeout 200 python tools/stream_pool_probe.py nccl_first bf16 40 2>&1 | grep "\[streams\]" >> $O
done
for v in "HN_HEAD_PRIO=0" "HN_STREAMS=cumask"; do
  echo "# env $v (clean)" >> $O
  env $v timeout 200 python tools/stream_pool_probe.py clean bf16 40 2>&1 | grep "\[streams\]" >> $O
done
cat $O
