#!/bin/bash
# Kernel-iteration call: gpurun WITHOUT the 49 MB config-5 fixtures (push time is charged to the GPU budget).
# usage: tools/gpu_iter.sh <timeout-seconds> '<command>'      (the ignore file is removed again on exit, whatever happens)
cd "$(dirname "$0")/.."
trap 'rm -f .gpurunignore' EXIT
printf 'tests/golden/config5/\nbuild/\n' > .gpurunignore
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
