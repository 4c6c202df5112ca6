#!/bin/bash
# round 4: the atomics-free EdgeConv backward - parity, then A/B against the atomic scatter on the configs[3] bench
set -e
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_edge_bwd.py tests/test_gpu_resgcn.py tests/test_gpu_resgcn28.py tests/test_gpu_resgcn_variants.py tests/test_gpu_modules.py -x -q -m gpu > $O/edge_bwd_tests.log 2>&1
tail -3 $O/edge_bwd_tests.log
timeout -k 10 200 python bench.py --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline > $O/gcn_gather.json 2> $O/gcn_gather.err
PSG_GCN_EDGE_BWD=atomic timeout -k 10 200 python bench.py --allow-env-switches --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline > $O/gcn_atomic.json 2> $O/gcn_atomic.err
python - <<'PY'
import json
for n in ("gather","atomic"):
    d=json.loads([l for l in open("gpurun_out/r04/gcn_%s.json"%n) if l.startswith("{")][-1])
    print(n, round(d["value"],3), d.get("kernel_ms_per_iteration"), d["config"].get("env_switches"))
PY
