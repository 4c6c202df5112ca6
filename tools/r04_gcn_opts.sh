#!/bin/bash
# configs[3] after the round-4 kNN kernel: fused [P | Q] product on / off, 3 / 4 launches in flight, coalescing 4 / 6 rooms
O=gpurun_out/r04; mkdir -p $O
run() { name=$1; shift
  timeout -k 10 200 python bench.py --allow-env-switches --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline --no-reference "$@" > $O/gcn_$name.json 2> $O/gcn_$name.err || { tail -5 $O/gcn_$name.err; return 1; }
  python -c "
import json
d=json.loads([l for l in open('$O/gcn_$name.json') if l.startswith('{')][-1])
print('$name', round(d['value'],3), 'rooms/s', d.get('kernel_ms_per_iteration'))"
}
run base && PSG_GCN_PQ_FUSION=1 run pq && run c4 --gcn-concurrency 4 && run c2 --gcn-concurrency 2 && run co6 --gcn-coalesce 6 --steps 36
