#!/bin/bash
# end-of-round verification on the final tree: build(), smoke(), the default bench command (-> profiles/r05_bench_final.json)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r5final_bench.json 2> gpurun_out/r5final_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5final_bench.json'))
print('headline', d['ms_per_step'], d['value'], d['step_launch'][:20], 'replay', d['replayed_steps'], 'eager', d['eager_launches'], 'streamed', d['streamed_batches']['ms_per_step'], 'peak', d['peak_hbm_GB'])
r=d['roofline']; print('roofline', r['achieved'], r['frac'], r['traffic'])
print({k:(v['ms_per_launch'], v['frac']) for k,v in r['in_step'].items() if isinstance(v,dict) and 'frac' in v})
for k,v in d.get('other_configs',{}).items(): print(k, v.get('ms_per_step'), v.get('eager_launches'), v.get('replayed_steps'), 'roofline', (v.get('roofline') or {}).get('frac'), 'cpu', (v.get('cpu_baseline') or {}).get('value'))
print('cpu', d['cpu_baseline']['value'])
PY
