#!/bin/bash
# round 5, step k: launch shape at the driver's --steps 20 --warmup 5 after the FP split (steps coalesced per launch x launches in flight)
mkdir -p gpurun_out/r5k2
for cfg in "8 3" "8 4" "5 4" "4 5" "7 3" "10 2" "4 3"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --coalesce $1 --concurrency $2 --no-secondary --no-reference --no-cpu-baseline > gpurun_out/r5k2/s20_c$1_n$2.json 2> gpurun_out/r5k2/s20_c$1_n$2.err
  python - "$1" "$2" <<'P'
import json,sys
d=json.loads(open("gpurun_out/r5k2/s20_c%s_n%s.json"%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print("steps20 coalesce",sys.argv[1],"in flight",sys.argv[2],"->",round(d["value"],1))
P
done
for cfg in "8 3" "8 4" "12 3"; do
  set -- $cfg
  python bench.py --coalesce $1 --concurrency $2 --no-secondary --no-reference --no-cpu-baseline > gpurun_out/r5k2/s48_c$1_n$2.json 2> gpurun_out/r5k2/s48_c$1_n$2.err
  python - "$1" "$2" <<'P'
import json,sys
d=json.loads(open("gpurun_out/r5k2/s48_c%s_n%s.json"%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print("steps48 coalesce",sys.argv[1],"in flight",sys.argv[2],"->",round(d["value"],1))
P
done
