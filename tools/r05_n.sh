#!/bin/bash
# round 5, step n: ResGCN launch shape (rooms coalesced per launch x launches in flight)
mkdir -p gpurun_out/r5n2
for cfg in "4 3 24" "4 4 32" "2 6 24" "8 2 32" "6 3 36" "4 5 40" "3 4 24"; do
  set -- $cfg
  python bench.py --workload resgcn --gcn-coalesce $1 --gcn-concurrency $2 --steps $3 --warmup $1 --no-cpu-baseline --no-reference > gpurun_out/r5n2/gcn_c$1_n$2.json 2> gpurun_out/r5n2/gcn_c$1_n$2.err
  python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/r5n2/gcn_c%s_n%s.json"%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
    print("resgcn coalesce",sys.argv[1],"in flight",sys.argv[2],"->",round(d["value"],2))
except Exception as e:
    print("resgcn", sys.argv[1], sys.argv[2], "failed", e)
P
done
