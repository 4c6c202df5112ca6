#!/bin/bash
# configs[3] with ONE launch in flight and with the default three: gather vs atomic EdgeConv backward
O=gpurun_out/r04; mkdir -p $O
for c in 1 3; do
for mode in gather atomic; do
  if [ $mode = atomic ]; then export PSG_GCN_EDGE_BWD=atomic; else unset PSG_GCN_EDGE_BWD; fi
  timeout -k 10 200 python bench.py --allow-env-switches --workload resgcn --steps 24 --warmup 8 --gcn-concurrency $c --no-cpu-baseline --no-reference > $O/gcn_${mode}_c$c.json 2> $O/gcn_${mode}_c$c.err || exit 1
  python -c "
import json
d=json.loads([l for l in open('$O/gcn_${mode}_c$c.json') if l.startswith('{')][-1])
print('$mode c=$c', round(d['value'],3), 'rooms/s', d.get('kernel_ms_per_iteration'))"
done; done
