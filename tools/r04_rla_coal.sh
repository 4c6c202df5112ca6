#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
for g in 8 16 24; do
timeout -k 10 400 python bench.py --workload randla --rla-coalesce $g --steps $((g*3)) --warmup $g --no-cpu-baseline --no-reference > $O/rla_g$g.json 2> $O/rla_g$g.err || { tail -5 $O/rla_g$g.err; exit 1; }
python -c "
import json
d=json.loads([l for l in open('$O/rla_g$g.json') if l.startswith('{')][-1])
print('coalesce $g', round(d['value'],2), d['unit'])"
done
