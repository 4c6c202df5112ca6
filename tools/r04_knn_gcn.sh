#!/bin/bash
# round 4: kNN / ResGCN parity tests, then configs[3] with the prefilter kernel serving dilations 1..MAXD
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_knn_bf16.py tests/test_gpu_knn_fused.py tests/test_gpu_resgcn28.py tests/test_gpu_resgcn.py tests/test_gpu_alt_paths.py -x -q > $O/knn_tests.log 2>&1 || { tail -30 $O/knn_tests.log; exit 1; }
tail -2 $O/knn_tests.log
for maxd in ${MAXDS:-3 12 20}; do
  PSG_GCN_KNN_BF_MAXD=$maxd timeout -k 10 200 python bench.py --allow-env-switches --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline --no-reference > $O/gcn_maxd$maxd.json 2> $O/gcn_maxd$maxd.err || { tail -5 $O/gcn_maxd$maxd.err; exit 1; }
  python -c "
import json
d=json.loads([l for l in open('$O/gcn_maxd$maxd.json') if l.startswith('{')][-1])
print('maxd=$maxd', round(d['value'],3), 'rooms/s', d.get('kernel_ms_per_iteration'), d.get('roofline'))"
done
