#!/bin/bash
# round 5, call C: does the one-rank RCCL process group cost the secondary lines their hardware-queue assignment?
mkdir -p gpurun_out/r5e
for w in pointnet2_msg tarnu; do
  PSG_BENCH_PG_FIRST=1 python bench.py --workload $w --steps 16 --warmup 8 --no-cpu-baseline --no-reference > gpurun_out/r5e/${w}_pgfirst.json 2> gpurun_out/r5e/${w}_pgfirst.err
  python bench.py --workload $w --steps 16 --warmup 8 --no-cpu-baseline --no-reference > gpurun_out/r5e/${w}_poolfirst.json 2> gpurun_out/r5e/${w}_poolfirst.err
  python bench.py --workload $w --steps 16 --warmup 8 --no-cpu-baseline --no-reference --no-process-group > gpurun_out/r5e/${w}_nopg.json 2> gpurun_out/r5e/${w}_nopg.err
done
python -m pytest tests/test_gpu_alt_paths.py -q -m gpu -k "PN2_SPLIT" 2>&1 | tail -3
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5e/*.json')):
    try: print(f, round(json.loads(open(f).read().strip().splitlines()[-1])['value'],1))
    except Exception as e: print(f,'ERR',e)
PY
