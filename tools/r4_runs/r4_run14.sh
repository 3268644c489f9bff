#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
bash tools/pp_ablate.sh run "d64_0 d64_32 d64_64 d64_128" layer1.x.conv2 8 > gpurun_out/r14_d64_ablation.txt 2>&1
cat gpurun_out/r14_d64_ablation.txt
