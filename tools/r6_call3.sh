#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_streams.txt; : > $O
for q in "" 8 16; do
  for m in clean pool_first pool_after nccl_first; do
    if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
    timeout 200 python tools/stream_pool_probe.py $m bf16 40 2>&1 | grep "\[streams\]" >> $O
  done
done
unset GPU_MAX_HW_QUEUES
for m in clean nccl_first; do timeout 200 python tools/stream_pool_probe.py $m f32 12 2>&1 | grep "\[streams\]" >> $O; done
cat $O
timeout 600 python tools/soak_determinism.py 150 0 > gpurun_out/r6_soak0.txt 2>&1; echo "soak0 rc $?"; grep "\[soak\]" gpurun_out/r6_soak0.txt | tail -4
timeout 600 python tools/soak_determinism.py 150 1 > gpurun_out/r6_soak1.txt 2>&1; echo "soak1 rc $?"; grep "\[soak\]" gpurun_out/r6_soak1.txt | tail -4
