#!/bin/bash
# configs[3], three launches in flight: the fused [P | Q] product of the next block (PSG_GCN_PQ_FUSION=1) against the default,
# with the atomics-free EdgeConv backward in both
O=gpurun_out/r04; mkdir -p $O
for rep in 1 2; do
for mode in default pq; do
  if [ $mode = pq ]; then export PSG_GCN_PQ_FUSION=1; else unset PSG_GCN_PQ_FUSION; fi
  timeout -k 10 200 python bench.py --allow-env-switches --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline --no-reference > $O/gcn_pq_${mode}_$rep.json 2> $O/gcn_pq_${mode}_$rep.err || exit 1
  python -c "
import json
d=json.loads([l for l in open('$O/gcn_pq_${mode}_$rep.json') if l.startswith('{')][-1])
print('$mode rep $rep', round(d['value'],3), 'rooms/s', d.get('kernel_ms_per_iteration'))"
done; done
