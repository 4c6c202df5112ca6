#!/bin/bash
# round 5, call F: RandLA-Net clouds per launch x launches in flight after the lfa16 / direct-GEMM changes
mkdir -p gpurun_out/r5j
for c in 8 16 24; do for k in 3 4; do
  python bench.py --workload randla --steps 48 --warmup 24 --rla-coalesce $c --concurrency $k --no-cpu-baseline --no-reference > gpurun_out/r5j/rla_c${c}_k${k}.json 2> gpurun_out/r5j/rla_c${c}_k${k}.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5j/rla_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],2), d['l_inf']['value'])
    except Exception as e: print(f,'ERR',e)
PY
