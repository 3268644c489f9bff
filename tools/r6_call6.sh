#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
except Exception as e:
    print('no line', e); sys.exit(0)
print('  f32', d['value'], d['roofline']['frac'], '| bf16', d.get('bf16_mode',{}).get('value'), d.get('bf16_mode',{}).get('pipelined',{}).get('value'), d.get('bf16_mode',{}).get('plain',{}).get('value'),
      '| train', d.get('train_bf16',{}).get('value'), '| layout', d.get('layout',{}).get('value'))
PY
}
for rep in 1 2; do
  for v in "HN_HEAD_PRIO=high" "HN_HEAD_PRIO=normal" "HN_BATCH_SIDE_STREAM=1"; do
    echo "# rep $rep env $v"
    env $v timeout 600 python bench.py --legs bf16,train,layout --no-cpu-baseline > gpurun_out/r6_ab_${rep}_${v%%=*}_${v##*=}.json 2>/dev/null
    summ gpurun_out/r6_ab_${rep}_${v%%=*}_${v##*=}.json
  done
done
echo "# force-rccl (normal prio), fp32 headline + legs"
timeout 600 python bench.py --force-rccl --legs bf16,train,layout --no-cpu-baseline > gpurun_out/r6_force_rccl.json 2>gpurun_out/r6_force_rccl.err; summ gpurun_out/r6_force_rccl.json
echo "# force-rccl (high prio)"
HN_HEAD_PRIO=high timeout 600 python bench.py --force-rccl --legs bf16,train,layout --no-cpu-baseline > gpurun_out/r6_force_rccl_high.json 2>/dev/null; summ gpurun_out/r6_force_rccl_high.json
