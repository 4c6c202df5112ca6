#!/bin/bash
# configs[3]: rooms per launch x launches in flight (the kNN grids are 128 workgroups per room: 4 rooms = 2 per CU)
O=gpurun_out/r04; mkdir -p $O
run() { name=$1; shift
  timeout -k 10 300 python bench.py --workload resgcn --warmup 8 --no-cpu-baseline --no-reference "$@" > $O/gcn_$name.json 2> $O/gcn_$name.err || { tail -5 $O/gcn_$name.err; return 1; }
  python -c "
import json
d=json.loads([l for l in open('$O/gcn_$name.json') if l.startswith('{')][-1])
print('$name', round(d['value'],3), 'rooms/s')"
}
run co4c3 --gcn-coalesce 4 --gcn-concurrency 3 --steps 24 && run co8c2 --gcn-coalesce 8 --gcn-concurrency 2 --steps 32 && run co8c3 --gcn-coalesce 8 --gcn-concurrency 3 --steps 48 && run co6c3 --gcn-coalesce 6 --gcn-concurrency 3 --steps 36 && run co12c2 --gcn-coalesce 12 --gcn-concurrency 2 --steps 48
