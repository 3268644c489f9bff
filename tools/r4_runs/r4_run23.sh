#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
PIPE_B=32 timeout 300 python tools/pipe_lanes.py bf16 20 2>&1 | grep lanes
PIPE_B=16 timeout 300 python tools/pipe_lanes.py bf16 40 2>&1 | grep lanes
