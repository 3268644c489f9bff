#!/bin/bash
# round 4, final evidence on ONE box: counter passes + kernel stats (forward bf16 / f32, training step), then the default bench line
# (which quotes those counters: same kernel sources -> traffic_stale false), then the whole GPU suite.   HN_GIT_HEAD=<commit> bash tools/r4_final.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/profile_forward.sh bf16 r4 > gpurun_out/r4_prof_bf16.log 2>&1
bash tools/profile_forward.sh f32 r4 > gpurun_out/r4_prof_f32.log 2>&1
python tools/merge_pmc.py gpurun_out r4 > gpurun_out/r4_pmc_forward.json 2> gpurun_out/r4_merge.err
bash tools/profile_train.sh r4 > gpurun_out/r4_prof_train.log 2>&1
cp gpurun_out/r4_pmc_forward.json gpurun_out/r4_pmc_train.json profiles/ 2>/dev/null
rm -rf gpurun_out/prof_r4_bf16 gpurun_out/prof_r4_f32
timeout 900 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
python - <<'PY'
import json
for l in open('gpurun_out/r4_bench_default.json'):
    if l.startswith('{'):
        d = json.loads(l)
        r = d['roofline']
        print('f32', d['value'], d['ms_per_step'], r['frac'], 'traffic', r.get('traffic'), 'stale', r.get('traffic_stale'))
        b = d.get('bf16_mode', {})
        print('bf16', b.get('value'), b.get('ms_per_step'), b.get('roofline', {}).get('frac'), b.get('steps'))
        t = d.get('train_bf16', {})
        print('train', t.get('value'), t.get('ms_per_step'), {k: t.get('roofline', {}).get(k) for k in ('frac', 'traffic', 'traffic_stale')})
        print('layout', d.get('layout', {}).get('value'), 'cpu', d.get('cpu_baseline', {}).get('value'), 'aug', d.get('augment_pipeline', {}).get('value'))
d = json.load(open('gpurun_out/r4_pmc_forward.json'))
for p, v in d['precisions'].items():
    print(p, 'total GB %.1f ratio %.2f' % (v['total_bytes'] / 1e9, v['counter_over_algorithmic']))
    for k, f in v['by_kernel_family'].items():
        if f.get('fetch_bytes', 0) + f.get('write_bytes', 0) > 2e8:
            print('   %-24s disp %5.1f fetch %6.2f GB write %6.2f GB mfma busy %s' % (k, f.get('dispatches_per_forward', 0), f.get('fetch_bytes', 0) / 1e9, f.get('write_bytes', 0) / 1e9, f.get('mfma_busy_pct')))
t = json.load(open('gpurun_out/r4_pmc_train.json'))
print('train step GB %.1f model %.1f ratio %.2f mfma busy %s' % (t['total_bytes'] / 1e9, t['model_bytes'] / 1e9, t['counter_over_model'], t['mfma_busy_pct_whole_step']))
PY
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r4_gpu_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r4_gpu_tests.txt | tail -3
