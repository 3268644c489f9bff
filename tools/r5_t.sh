#!/bin/bash
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|rror" | tail -5 > gpurun_out/t_parity.txt
timeout 200 python bench.py --steps 5 --warmup 2 --legs latency 2>/dev/null | tail -1 > gpurun_out/bench_lat.json
