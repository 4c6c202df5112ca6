#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-secondary --no-reference > $O/sw_$name.json 2> $O/sw_$name.err || { tail -3 $O/sw_$name.err; return 1; }
  python -c "
import json
d=json.loads([l for l in open('$O/sw_$name.json') if l.startswith('{')][-1])
print('$name', round(d['value'],2), d['unit'])"; }
run h_c8x3 --steps 48 --warmup 8
run h_c16x3 --steps 96 --warmup 16 --coalesce 16
run h_c8x4 --steps 64 --warmup 8 --concurrency 4
run h_c12x3 --steps 72 --warmup 12 --coalesce 12
run g_c4x3 --workload resgcn --steps 24 --warmup 8
run g_c6x3 --workload resgcn --steps 36 --warmup 12 --gcn-coalesce 6
run g_c8x2 --workload resgcn --steps 32 --warmup 8 --gcn-coalesce 8 --gcn-concurrency 2
run g_c4x4 --workload resgcn --steps 32 --warmup 8 --gcn-concurrency 4
