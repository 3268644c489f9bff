#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_streams2.txt; : > $O
for m in clean nccl_first nccl_init_only nccl_destroyed nccl_after gloo_first used1 used4 usedhi; do
  timeout 200 python tools/stream_pool_probe.py $m bf16 40 2>&1 | grep "\[streams\]" >> $O
done
for v in "NCCL_MAX_NCHANNELS=1" "HSA_NO_SCRATCH_RECLAIM=1" "GPU_MAX_HW_QUEUES=2" "RCCL_MSCCL_ENABLE=0" "HIP_FORCE_DEV_KERNARG=1"; do
  echo "# env $v" >> $O
  env $v timeout 200 python tools/stream_pool_probe.py nccl_first bf16 40 2>&1 | grep "\[streams\]" >> $O
done
cat $O
# f32 branch stream A/B on this box
for v in "HN_F32_BRANCH=0" "HN_F32_DEFER_JOIN=0" "HN_X=1"; do
  echo "# env $v"; env $v timeout 200 python tools/stream_pool_probe.py clean f32 12 2>&1 | grep "\[streams\]"
done
cd /tmp
for m in clean nccl_first; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o p -- python $GRAFT_REPO_ROOT/tools/stream_pool_probe.py $m bf16 20 > /tmp/prof_$m.log 2>&1
  f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1)
  echo "== $m $f"; head -12 "$f" | cut -c1-160
  python - "$f" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])): tot+=float(r.get('TotalDurationNs',0))
print('sum of kernel durations ms', tot/1e6)
PY
done
