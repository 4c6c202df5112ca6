#!/bin/bash
# round 5, call G: the whole GPU suite, then the default bench line
mkdir -p gpurun_out/r5l
python -m pytest tests -q -m gpu > gpurun_out/r5l/tests.txt 2>&1
tail -8 gpurun_out/r5l/tests.txt
python bench.py > gpurun_out/r5l/bench_default.json 2> gpurun_out/r5l/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5l/bench_default.json').read().strip().splitlines()[-1])
print('headline', round(d['value'],1), 'unco', round(d['uncoalesced_reference']['value'],1), 'api', round(d['api_level']['value'],1), 'scene', round(d['whole_scene']['value'],1), d['roofline']['frac'])
for k,v in d['secondary'].items():
    if 'error' in v: print(k, v['error']); continue
    print(k, round(v['value'],2), v['unit'], v.get('uncoalesced_reference',{}).get('value'), v.get('batch32_quirk',{}).get('value'), v['config'].get('hipgraph'), v.get('roofline',{}).get('kernel','')[:40], v.get('roofline',{}).get('frac'), v.get('bench_wall_s'))
PY
