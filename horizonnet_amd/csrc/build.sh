#!/bin/bash
# Build libhorizonnet_hip.so in-tree for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=${HN_BUILD_OUT:-../libhorizonnet_hip.so}      # HN_BUILD_OUT / HN_BUILD_FLAGS / HN_BUILD_OBJ: throw-away measurement builds (tools/ab_lib.sh)
OBJDIR=${HN_BUILD_OBJ:-../../build/obj}
SRCS="engine.hip engine_bf16.hip train.hip conv_igemm_f32.hip conv_wgrad_f32.hip conv_wgrad_bf16.hip bn_fold.hip conv_igemm_bf16.hip conv_igemm_bf16_pp.hip conv3x3_dwr_bf16.hip conv3x3_dwr64_bf16.hip stem_pool_bf16.hip elementwise.hip train_ops.hip lstm.hip lstm_wide_f32.hip lstm_bf16.hip panostretch.hip augment.hip labels.hip peaks.hip multi_job.hip layout_fit.hip"
mkdir -p $OBJDIR
OBJS=""
for f in $SRCS; do
  o=$OBJDIR/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ hn_common.h -nt "$o" ] || [ engine_internal.h -nt "$o" ] || [ multi_job.h -nt "$o" ] || [ conv_bf16_args.h -nt "$o" ] || [ conv_bf16_pp.h -nt "$o" ] || [ stat_commit.h -nt "$o" ] || [ ../../include/horizonnet_hip.h -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed ${HN_BUILD_FLAGS:-} -c "$f" -o "$o"
  fi
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $(readlink -f $OUT)"
# an unresolved kernel stub only shows up at dlopen time: fail the build, not the first GPU call
python3 -c "import ctypes,sys; ctypes.CDLL('$(readlink -f $OUT)')" || { echo "libhorizonnet_hip.so does not load"; exit 1; }
