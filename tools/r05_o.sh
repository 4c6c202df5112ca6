#!/bin/bash
# round 5, step o: launches in flight for the lockstep NU workload (PSG_BENCH_NU_LOCK_CONC) and for ResGCN
mkdir -p gpurun_out/r5o2
for cfg in "3 2" "4 2" "5 2" "4 3" "3 3"; do
  set -- $cfg
  PSG_BENCH_NU_LOCK_CONC=$1 python bench.py --workload tarnu --nu-coalesce $2 --no-cpu-baseline --no-reference > gpurun_out/r5o2/nu_$1_$2.json 2> gpurun_out/r5o2/nu_$1_$2.err
  python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/r5o2/nu_%s_%s.json"%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1]); print("tarnu in flight",sys.argv[1],"steps per call",sys.argv[2],"->",round(d["value"],1))
except Exception as e: print("tarnu", sys.argv[1:], "failed", e)
P
done
for c in 3 4; do
  python bench.py --workload resgcn --gcn-concurrency $c --no-cpu-baseline --no-reference > gpurun_out/r5o2/gcn_$c.json 2> gpurun_out/r5o2/gcn_$c.err
  python - "$c" <<'P'
import json,sys
d=json.loads(open("gpurun_out/r5o2/gcn_%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print("resgcn in flight",sys.argv[1],"->",round(d["value"],2))
P
done
