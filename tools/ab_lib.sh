#!/bin/bash
# A/B of two builds of libhorizonnet_hip.so on ONE GPU box (boxes differ by a few percent, so a before/after across two
# gpurun calls proves nothing).  tools/ab_lib.sh save a|b   -> copy the current build to tools/probe/ab_<x>.so (here);
# tools/ab_lib.sh run REPS '<command printing one number line>'   -> alternate a, b, a, b ... on the box, print every line.
cd "$(dirname "$0")/.."
LIB=horizonnet_amd/libhorizonnet_hip.so
if [ "$1" = "save" ]; then cp $LIB tools/probe/ab_$2.so; echo "saved tools/probe/ab_$2.so"; exit 0; fi
REPS=$2; CMD=$3
cp $LIB /tmp/ab_orig.so
for r in $(seq 1 $REPS); do
  for v in a b; do
    cp tools/probe/ab_$v.so $LIB
    echo "[$v] $(bash -c "$CMD" 2>/dev/null | tail -${AB_TAIL:-1} | tr '\n' ' ')"
  done
done
cp /tmp/ab_orig.so $LIB
