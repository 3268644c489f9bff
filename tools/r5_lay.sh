#!/bin/bash
O=$(pwd)/gpurun_out
rm -f $O/lay.txt
for legs in train,layout; do for m in 1 0; do
HN_BATCH_SIDE_STREAM=$m timeout 300 python bench.py --steps 5 --warmup 2 --legs $legs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('legs=$legs side=$m layout', r['layout']['value'], r['layout']['ms_per_step'], 'train', r.get('train_bf16',{}).get('value'))" >> $O/lay.txt
done; done
