#!/bin/bash
# Measurement builds of the library with parts of the ping-pong conv loop switched off (-DHN_PP_ABL=mask: 1 no LDS-DMA, 2 no
# fragment reads, 4 no barriers, 8 no MFMAs, 16 no counted waits; results are WRONG, only the timing means something) or with
# another MFMA order (-DHN_PP_MMORD).  Only conv_igemm_bf16_pp.hip is recompiled.  usage: tools/pp_ablate.sh build "0 1 2 4 ..."
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  mkdir -p tools/probe
  for A in $2; do
    rm -rf build/obj_abl && cp -r build/obj build/obj_abl && rm -f build/obj_abl/conv_igemm_bf16_pp.o
    if [ "${A:0:4}" = "d64_" ]; then FL="-DHN_D64_ABL=${A:4}"; rm -f build/obj_abl/conv3x3_dwr64_bf16.o; elif [ "$A" = "ord" ]; then FL="-DHN_PP_MMORD"; elif [ "$A" = "stamp1" ] || [ "$A" = "stamp2" ]; then FL="-DHN_PP_STAMP=${A#stamp} -DHN_CONV_TRACE"; rm -f build/obj_abl/conv_igemm_bf16.o; elif [ "$A" = "lgkb" ]; then FL="-DHN_PP_LGKB"; elif [ "${A:0:3}" = "pol" ]; then FL="-DHN_PP_POLICY_A=${A:3:1} -DHN_PP_POLICY_B=${A:4:1}"; else FL="-DHN_PP_ABL=$A"; fi
    HN_BUILD_OUT=$PWD/tools/probe/pp_abl_$A.so HN_BUILD_OBJ=$PWD/build/obj_abl HN_BUILD_FLAGS="$FL" bash horizonnet_amd/csrc/build.sh 2>&1 | grep -v warning | tail -1
  done
  rm -rf build/obj_abl
  exit 0
fi
# on the GPU box: tools/pp_ablate.sh run "0 1 2 ..." SHAPES
for A in $2; do
  echo "## ablation $A"
  SWEEP_LIB=tools/probe/pp_abl_$A.so SWEEP_NOASSERT=1 SWEEP_ONLY=$3 SWEEP_VARIANTS=${4:-4} timeout -k 5 120 python tools/conv_sweep.py 2>&1 | grep -v "^#\|MISMATCH\|amdgpu.ids"
done
