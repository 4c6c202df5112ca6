#!/bin/bash
# round 5, call D: the prefilter kNN kernel with the unit-balanced final ranking: parity, timeline, ResGCN bench; harness prefetch
mkdir -p gpurun_out/r5f
python -m pytest tests/test_gpu_knn_bf16.py tests/test_gpu_knn_fused.py tests/test_gpu_resgcn28.py tests/test_gpu_resgcn.py tests/test_gpu_harness.py -q -m gpu -x > gpurun_out/r5f/tests.txt 2>&1
tail -5 gpurun_out/r5f/tests.txt
KNN_D=1,6,12,20,24 PSG_GCN_KNN=bf16 PSG_LIBRARY_OVERRIDE=build/libpsg_tl.so python tools/knn_timeline.py 4 > gpurun_out/r5f/tl_bf16.txt 2>&1
tail -6 gpurun_out/r5f/tl_bf16.txt
python bench.py --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline > gpurun_out/r5f/bench_gcn.json 2> gpurun_out/r5f/bench_gcn.err
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5f/bench_pn2.json 2> gpurun_out/r5f/bench_pn2.err
python - <<'PY'
import json
for n in ('gcn','pn2'):
    try:
        d=json.loads(open('gpurun_out/r5f/bench_%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d['value'],2), d.get('whole_scene',{}).get('value'), d.get('api_level',{}).get('value'), d.get('roofline',{}).get('frac'))
    except Exception as e: print(n,'ERR',e)
PY
