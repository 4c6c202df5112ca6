#!/bin/bash
# round 6, call G: the NU loop's geometry plan horizon (forwards per plan) against launches in flight
mkdir -p gpurun_out/r6l
for conc in 4 1; do for pa in 10 20 30 50; do
  python bench.py --workload tarnu --no-cpu-baseline --no-reference --nu-plan-ahead $pa --nu-concurrency $conc > gpurun_out/r6l/t_c${conc}_p$pa.json 2> gpurun_out/r6l/t_c${conc}_p$pa.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6l/t_c${conc}_p$pa.json').read().strip().splitlines()[-1]); k=d['kernel_ms_per_attack']
print('in flight $conc plan_ahead $pa:', round(d['value'],1), 'geometry ms', round(k['fps']+k['three_nn']+k['ball_query'],1), d['config'].get('hipgraph'))
PY
done; done
