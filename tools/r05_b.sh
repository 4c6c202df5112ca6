#!/bin/bash
# round 5, call B: parity tests touched by the SA split / harness streams, then the default bench line
mkdir -p gpurun_out/r5d
python -m pytest tests/test_gpu_harness.py tests/test_gpu_nu.py tests/test_gpu_msg.py tests/test_gpu_alt_paths.py tests/test_gpu_resgcn28.py -q -m gpu -k "not RLA and not GCN" > gpurun_out/r5d/tests.txt 2>&1
tail -6 gpurun_out/r5d/tests.txt
python bench.py > gpurun_out/r5d/bench_default.json 2> gpurun_out/r5d/bench_default.err
tail -c 300 gpurun_out/r5d/bench_default.err
