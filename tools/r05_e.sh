#!/bin/bash
# round 5, call E: MSG split + XCD-ordered row GEMM: parity, then the affected bench lines
mkdir -p gpurun_out/r5g
python -m pytest tests/test_gpu_msg.py tests/test_gpu_harness.py tests/test_gpu_modules.py tests/test_gpu_resgcn.py tests/test_gpu_resgcn28.py tests/test_gpu_resgcn_variants.py tests/test_randla_net.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r5g/tests.txt 2>&1
tail -5 gpurun_out/r5g/tests.txt
for w in pointnet2_msg resgcn randla; do
  python bench.py --workload $w --no-cpu-baseline --no-reference > gpurun_out/r5g/bench_$w.json 2> gpurun_out/r5g/bench_$w.err
done
python - <<'PY'
import json
for n in ('pointnet2_msg','resgcn','randla'):
    try:
        d=json.loads(open('gpurun_out/r5g/bench_%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d['value'],2), d.get('roofline',{}).get('frac'), d.get('kernel_ms_per_attack') or d.get('profiled_kernels_ms_per_iteration'))
    except Exception as e: print(n,'ERR',e)
PY
