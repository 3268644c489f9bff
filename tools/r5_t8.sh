#!/bin/bash
O=$(pwd)/gpurun_out
for i in 1 2 3; do
timeout 600 python bench.py --gpus 8 --backend gloo --share-gpu --mode train --dtype bf16 --allreduce-dtype bf16 --batch 1 --steps 2 --warmup 1 --rooms 4 > $O/t8_$i.out 2> $O/t8_$i.err
echo "run $i rc=$?" >> $O/t8_summary.txt
grep -v "Gloo\|Warning\|warn" $O/t8_$i.err | tail -30 > $O/t8_$i.tail
done
