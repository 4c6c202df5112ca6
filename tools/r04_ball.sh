#!/bin/bash
# headline bench with the grid ball query (default) and with the full scan (PSG_BALL_QUERY=scan)
O=gpurun_out/r04; mkdir -p $O
for rep in 1; do for mode in grid; do
  if [ $mode = scan ]; then export PSG_BALL_QUERY=scan; else unset PSG_BALL_QUERY; fi
  timeout -k 10 300 python bench.py --allow-env-switches --steps 48 --warmup 8 --no-cpu-baseline --no-secondary --no-reference > $O/ball_${mode}_$rep.json 2> $O/ball_${mode}_$rep.err || { tail -3 $O/ball_${mode}_$rep.err; exit 1; }
  python -c "
import json
d=json.loads([l for l in open('$O/ball_${mode}_$rep.json') if l.startswith('{')][-1])
k=d['kernel_ms_per_attack']
print('$mode rep $rep', round(d['value'],1), 'rooms/s  ball_query', k.get('ball_query'), 'fps', k.get('fps'), 'three_nn', k.get('three_nn'), 'total', d['kernel_ms_total_per_attack'])"
done; done
