#!/bin/bash
# round 6, call E: direct per-vertex GEMM of the EdgeConv layers: parity, A/B on one box, kernel stats
mkdir -p gpurun_out/r6i
python -m pytest tests/test_gpu_resgcn28.py tests/test_gpu_resgcn.py tests/test_gpu_resgcn_variants.py tests/test_gpu_edge.py tests/test_gpu_edge_bwd.py tests/test_gpu_modules.py -x -q -m gpu > gpurun_out/r6i/tests.log 2>&1 || { tail -30 gpurun_out/r6i/tests.log; exit 1; }
tail -2 gpurun_out/r6i/tests.log
for rep in 1 2; do for v in 1 0; do
  PSG_GCN_DIRECT_GEMM=$v python bench.py --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline --no-reference --allow-env-switches > gpurun_out/r6i/gcn_d$v.json 2> gpurun_out/r6i/gcn_d$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6i/gcn_d$v.json').read().strip().splitlines()[-1]); print('direct=$v', round(d['value'],3))
PY
done; done
O=gpurun_out/r6i ROWS=10 bash tools/gcn_stats.sh
