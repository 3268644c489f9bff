#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export HN_GIT_HEAD=$(cat gpurun_out/.git_head 2>/dev/null || echo unknown)
bash tools/profile_forward.sh bf16 r4 > gpurun_out/r4_prof_bf16.log 2>&1
bash tools/profile_forward.sh f32 r4 > gpurun_out/r4_prof_f32.log 2>&1
python tools/merge_pmc.py gpurun_out r4 > gpurun_out/r4_pmc_forward.json 2> gpurun_out/r4_merge.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_pmc_forward.json'))
for p,v in d['precisions'].items():
    print(p, 'total GB %.1f ratio %.2f' % (v['total_bytes']/1e9, v['counter_over_algorithmic']))
    for k,f in v['by_kernel_family'].items():
        print('   %-24s disp %5.1f fetch %6.2f GB write %6.2f GB mfma busy %s' % (k, f.get('dispatches_per_forward',0), f.get('fetch_bytes',0)/1e9, f.get('write_bytes',0)/1e9, f.get('mfma_busy_pct')))
print(d.get('measured_on'))
PY
rm -rf gpurun_out/prof_r4_bf16 gpurun_out/prof_r4_f32
