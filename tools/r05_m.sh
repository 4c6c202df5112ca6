#!/bin/bash
# round 5, step m: RandLA-Net launch shape (clouds coalesced per launch x launches in flight)
mkdir -p gpurun_out/r5m
for cfg in "16 3 48" "24 2 48" "24 3 72" "32 2 64" "12 4 48" "16 4 64"; do
  set -- $cfg
  python bench.py --workload randla --rla-coalesce $1 --concurrency $2 --steps $3 --warmup $1 --no-cpu-baseline --no-reference > gpurun_out/r5m/rla_c$1_n$2.json 2> gpurun_out/r5m/rla_c$1_n$2.err
  python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.loads(open("gpurun_out/r5m/rla_c%s_n%s.json"%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
    print("randla coalesce",sys.argv[1],"in flight",sys.argv[2],"->",round(d["value"],2))
except Exception as e:
    print("randla", sys.argv[1], sys.argv[2], "failed", e)
P
done
