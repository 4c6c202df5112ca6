#!/bin/bash
# round 6, call H: configs[3] launch shape, three repetitions each on one box (rooms per launch x launches in flight)
mkdir -p gpurun_out/r6m
for rep in 1 2 3; do for cfg in "4 4" "8 3" "6 4" "8 4" "4 6"; do set -- $cfg
  v=$(python bench.py --workload resgcn --steps 48 --warmup 8 --no-cpu-baseline --no-reference --gcn-coalesce $1 --gcn-concurrency $2 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],3))")
  echo "rep $rep coalesce $1 in-flight $2: $v"
done; done
