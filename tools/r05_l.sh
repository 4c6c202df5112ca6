#!/bin/bash
# round 5, step l: fp_fwd at five waves per SIMD (96 VGPRs) - parity tests of the PointNet++ paths, then the headline bench
set -o pipefail
mkdir -p gpurun_out/r5l
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_msg.py tests/test_gpu_edge.py tests/test_gpu_nu.py tests/test_gpu_api.py tests/test_gpu_harness.py -x -q -m gpu > gpurun_out/r5l/tests.txt 2>&1
rc=$?
tail -3 gpurun_out/r5l/tests.txt
[ $rc -ne 0 ] && exit $rc
python bench.py --no-secondary --no-reference --no-cpu-baseline > gpurun_out/r5l/bench.json 2> gpurun_out/r5l/bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r5l/bench.json").read().strip().splitlines()[-1])
print(round(d["value"],1), d["kernel_ms_per_attack"])
P
