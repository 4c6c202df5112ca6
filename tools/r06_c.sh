#!/bin/bash
# round 6, call C: configs[3] launch-shape / fusion switches re-measured after the kNN kernel moved
mkdir -p gpurun_out/r6d
run() { name=$1; shift; "$@" > gpurun_out/r6d/$name.json 2> gpurun_out/r6d/$name.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6d/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],3))
except Exception as e: print('$name ERR', e)
PY
}
B="python bench.py --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline --no-reference"
run base $B
PSG_GCN_PQ_FUSION=1 run pq $B --allow-env-switches
run c3 $B --gcn-concurrency 3
run c5 $B --gcn-concurrency 5
run co6 $B --gcn-coalesce 6
run co8c3 $B --gcn-coalesce 8 --gcn-concurrency 3
run base2 $B
