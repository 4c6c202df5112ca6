#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
for qc in "16 16" "24 24" "32 32" "32 16"; do set -- $qc; q=$1; c=$2
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --workload tarnu --nu-mode per-room-calls --nu-concurrency $c --steps 4 --warmup 2 --no-cpu-baseline > $O/tarnu_calls_q${q}_c$c.json 2> $O/tarnu_calls_q${q}_c$c.err || { tail -5 $O/tarnu_calls_q${q}_c$c.err; exit 1; }
  python -c "
import json
d=json.loads([l for l in open('$O/tarnu_calls_q${q}_c$c.json') if l.startswith('{')][-1])
print('queues $q conc $c', round(d['value'],1), d['unit'], d['optimizer_steps_per_attack'])"
done
for q in 4 16; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-secondary --no-reference > $O/head_q$q.json 2> $O/head_q$q.err || exit 1
python -c "
import json
d=json.loads([l for l in open('$O/head_q$q.json') if l.startswith('{')][-1])
print('headline queues $q', round(d['value'],1))"
done
