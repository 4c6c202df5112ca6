#!/bin/bash
# round 5, call I: the row GEMM's tile order A/B on ONE box (kernel stats of the ResGCN workload, then the bench line)
export TMPDIR=/tmp
mkdir -p gpurun_out/r5o
for v in new legacy new legacy; do
  if [ $v = legacy ]; then export PSG_LIBRARY_OVERRIDE=build/libpsg_legacy.so; else unset PSG_LIBRARY_OVERRIDE; fi
  n=$(ls gpurun_out/r5o | grep -c "^${v}_.*csv")
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5o/tmp -o s -- python3 bench.py --allow-env-switches --no-process-group --workload resgcn --steps 4 --warmup 4 --gcn-concurrency 1 --no-cpu-baseline --no-reference > gpurun_out/r5o/${v}_$n.log 2>&1
  cp $(find gpurun_out/r5o/tmp -name '*kernel_stats.csv' | head -1) gpurun_out/r5o/${v}_$n.csv; rm -rf gpurun_out/r5o/tmp
  echo $v $(grep "gemm_rows_kernel<2, 2, 1, false, 2, 2>" gpurun_out/r5o/${v}_$n.csv | cut -d, -f2-4) $(grep "gemm_rows_kernel<2, 2, 0, false, 2, 2>" gpurun_out/r5o/${v}_$n.csv | cut -d, -f4)
  python bench.py --allow-env-switches --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline --no-reference 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench', round(d['value'],2))"
done
