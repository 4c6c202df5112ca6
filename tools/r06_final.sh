#!/bin/bash
# round 6, final: PART 1 = the whole GPU suite, the default bench line, the driver's command, every workload alone;
# PART 2 = the profiler passes behind profiles/r06_* (tools/profile_round.sh r06 all).  One gpurun call each (20-minute limit).
mkdir -p gpurun_out/r6z
if [ "${1:-1}" = 1 ]; then
python -m pytest tests -q -m gpu > gpurun_out/r6z/tests.txt 2>&1
tail -3 gpurun_out/r6z/tests.txt
python bench.py > gpurun_out/r6z/bench_default.json 2> gpurun_out/r6z/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > gpurun_out/r6z/bench_steps20.json 2> gpurun_out/r6z/bench_steps20.err
for w in tarnu resgcn pointnet2_msg randla; do python bench.py --workload $w > gpurun_out/r6z/bench_$w.json 2> gpurun_out/r6z/bench_$w.err; done
python - <<'P'
import json
for n in ("default","steps20","tarnu","resgcn","pointnet2_msg","randla"):
    try:
        d=json.loads(open("gpurun_out/r6z/bench_%s.json"%n).read().strip().splitlines()[-1]); print(n, round(d["value"],2), d["unit"])
    except Exception as e: print(n, "failed", e)
P
else
bash tools/profile_round.sh r06 all > gpurun_out/r6z/profile.log 2>&1
tail -2 gpurun_out/r6z/profile.log
ls gpurun_out/prof_r06
fi
