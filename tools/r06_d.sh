#!/bin/bash
# round 6, call D: headline A/B on one box: round-5 library vs current (readfirstlane builtin + mfma_fence)
mkdir -p gpurun_out/r6g
for rep in 1 2; do for lib in new old; do
  if [ $lib = old ]; then export PSG_LIBRARY_OVERRIDE=build/libpsg_r5.so; sw=--allow-env-switches; else unset PSG_LIBRARY_OVERRIDE; sw=; fi
  python bench.py --no-secondary --no-cpu-baseline --no-reference --steps 20 --warmup 5 $sw > gpurun_out/r6g/pn2_${lib}_$rep.json 2> gpurun_out/r6g/pn2_${lib}_$rep.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6g/pn2_${lib}_$rep.json').read().strip().splitlines()[-1]); print('$lib $rep', round(d['value'],1), d['roofline']['frac'])
PY
done; done
