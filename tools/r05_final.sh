#!/bin/bash
# round 5, final: the whole GPU suite, the default bench line, then the profiler passes behind profiles/r05_*
mkdir -p gpurun_out/r5z
python -m pytest tests -q -m gpu > gpurun_out/r5z/tests.txt 2>&1
tail -3 gpurun_out/r5z/tests.txt
python bench.py > gpurun_out/r5z/bench_default.json 2> gpurun_out/r5z/bench_default.err
python bench.py --steps 20 --warmup 8 --no-secondary > gpurun_out/r5z/bench_steps20.json 2> gpurun_out/r5z/bench_steps20.err
for w in tarnu resgcn pointnet2_msg randla; do python bench.py --workload $w > gpurun_out/r5z/bench_$w.json 2> gpurun_out/r5z/bench_$w.err; done
bash tools/profile_round.sh r05 all > gpurun_out/r5z/profile.log 2>&1
tail -2 gpurun_out/r5z/profile.log
