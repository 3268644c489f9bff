#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/poison_hunt.py 1 1 > gpurun_out/r6_hunt3_1_1.txt 2>&1; echo "hunt rc $?"; grep "\[hunt\]" gpurun_out/r6_hunt3_1_1.txt | tail -12
