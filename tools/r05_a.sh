#!/bin/bash
# round 5, call A: parity tests touched by the SA split / harness streams, SA1 forward-split A/B, kNN phase timelines
mkdir -p gpurun_out/r5c
python -m pytest tests/test_gpu_harness.py tests/test_gpu_nu.py tests/test_gpu_msg.py tests/test_gpu_alt_paths.py -x -q -m gpu -k "not RLA and not GCN" > gpurun_out/r5c/tests.txt 2>&1
tail -4 gpurun_out/r5c/tests.txt
PSG_PN2_SPLIT=2 python bench.py --no-secondary --no-cpu-baseline --no-reference --allow-env-switches > gpurun_out/r5c/bench_split2.json 2> gpurun_out/r5c/bench_split2.err
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5c/bench_split1.json 2> gpurun_out/r5c/bench_split1.err
for kind in bf16 f32; do
  KNN_D=1,6,12,20,24,27 PSG_GCN_KNN=$kind PSG_LIBRARY_OVERRIDE=build/libpsg_tl.so python tools/knn_timeline.py 4 > gpurun_out/r5c/tl_$kind.txt 2>&1
done
KNN_D=1,6,12,20,27 PSG_GCN_KNN=bf16 PSG_LIBRARY_OVERRIDE=build/libpsg_ph.so python tools/knn_timeline.py 4 > gpurun_out/r5c/ph_bf16.txt 2>&1
cat gpurun_out/r5c/tl_bf16.txt | tail -8
