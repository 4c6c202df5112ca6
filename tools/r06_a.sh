#!/bin/bash
# round 6, call A: prefilter kNN kernel with the small products accumulated first (A = 2^-14, B = 2^-19) and quad-fetched
# finalist rows: parity tests, then the kernel on the network's own features and the configs[3] line, new library against
# the round-5 one (build/libpsg_r5.so) on the same box
mkdir -p gpurun_out/r6a
python -m pytest tests/test_gpu_knn_bf16.py tests/test_gpu_knn_fused.py tests/test_gpu_resgcn28.py tests/test_gpu_resgcn.py -x -q -m gpu > gpurun_out/r6a/tests.log 2>&1 || { tail -30 gpurun_out/r6a/tests.log; exit 1; }
tail -3 gpurun_out/r6a/tests.log
BLOCKS=1,3,6,9,12,16,20,22,24,27 python tools/knn_real_feats.py > gpurun_out/r6a/real_new.log 2>&1 || { tail -20 gpurun_out/r6a/real_new.log; exit 1; }
BLOCKS=1,3,6,9,12,16,20,22,24,27 PSG_LIBRARY_OVERRIDE=build/libpsg_r5.so python tools/knn_real_feats.py > gpurun_out/r6a/real_old.log 2>&1
grep -h "^block" gpurun_out/r6a/real_new.log | cut -c1-20,100-400
echo ---- old
grep -h "^block" gpurun_out/r6a/real_old.log | cut -c1-20,100-400
for lib in new old new old; do
  if [ $lib = old ]; then export PSG_LIBRARY_OVERRIDE=build/libpsg_r5.so; sw=--allow-env-switches; else unset PSG_LIBRARY_OVERRIDE; sw=; fi
  python bench.py --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline --no-reference $sw > gpurun_out/r6a/gcn_$lib.json 2> gpurun_out/r6a/gcn_$lib.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6a/gcn_$lib.json').read().strip().splitlines()[-1]); print('$lib', d['value'], d.get('roofline',{}).get('frac'))
PY
done
