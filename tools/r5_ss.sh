#!/bin/bash
O=$(pwd)/gpurun_out
timeout 400 python -m pytest tests/test_gpu_dataset.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -4 > $O/ss_test.txt
for m in 1 0 1 0; do HN_BATCH_SIDE_STREAM=$m timeout 200 python bench.py --mode train --dtype bf16 --batch 64 --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('side=$m', r['value'], r['ms_per_step'], r.get('final_loss'))"; done >> $O/ss_test.txt
