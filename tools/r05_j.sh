#!/bin/bash
# round 5, step j: FP split (interpolated part of fp1-fp3's first layer per coarse point): parity, then A/B bench
set -o pipefail
mkdir -p gpurun_out/r5j
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_msg.py tests/test_gpu_edge.py tests/test_gpu_nu.py -x -q -m gpu > gpurun_out/r5j/tests.txt 2>&1
rc=$?
tail -25 gpurun_out/r5j/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-secondary > gpurun_out/r5j/bench_on.json 2> gpurun_out/r5j/bench_on.err && tail -c 600 gpurun_out/r5j/bench_on.json &&
PSG_PN2_FPSPLIT=0 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-secondary --allow-env-switches > gpurun_out/r5j/bench_off.json 2> gpurun_out/r5j/bench_off.err && tail -c 600 gpurun_out/r5j/bench_off.json
