#!/bin/bash
# configs[3] with the prefilter kernel serving dilations 1..MAXD (after the round-4 kernel)
O=gpurun_out/r04; mkdir -p $O
for maxd in ${MAXDS:-20 22 24 27}; do
  PSG_GCN_KNN_BF_MAXD=$maxd timeout -k 10 200 python bench.py --allow-env-switches --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline --no-reference > $O/gcn_maxd$maxd.json 2> $O/gcn_maxd$maxd.err || { tail -5 $O/gcn_maxd$maxd.err; exit 1; }
  python -c "
import json
d=json.loads([l for l in open('$O/gcn_maxd$maxd.json') if l.startswith('{')][-1])
print('maxd=$maxd', round(d['value'],3), 'rooms/s', d.get('kernel_ms_per_iteration'))"
done
