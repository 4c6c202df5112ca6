#!/bin/bash
# round 6, call B: timeline / phases of the prefilter kernel after the 2048-bin final ranking; the kernel beyond d = 20
mkdir -p gpurun_out/r6b
KNN_D=1,6,12,20,24,27 PSG_GCN_KNN=bf16 PSG_LIBRARY_OVERRIDE=build/libpsg_tl.so python tools/knn_timeline.py 4 > gpurun_out/r6b/tl_bf16.txt 2>&1
cat gpurun_out/r6b/tl_bf16.txt | cut -c1-330
KNN_D=1,6,12,20,24,27 PSG_GCN_KNN=bf16 PSG_LIBRARY_OVERRIDE=build/libpsg_ph.so python tools/knn_timeline.py 4 > gpurun_out/r6b/ph_bf16.txt 2>&1
grep phases gpurun_out/r6b/ph_bf16.txt | cut -c1-330
BLOCKS=20,21,22,23,24,25,26,27 PSG_GCN_KNN_BF_MAXD=27 python tools/knn_real_feats.py > gpurun_out/r6b/real_maxd27.log 2>&1
grep -h "^block" gpurun_out/r6b/real_maxd27.log | cut -c1-20,100-400
for md in 20 23 25 27; do
  PSG_GCN_KNN_BF_MAXD=$md python bench.py --workload resgcn --steps 24 --warmup 8 --no-cpu-baseline --no-reference --allow-env-switches > gpurun_out/r6b/gcn_md$md.json 2> gpurun_out/r6b/gcn_md$md.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6b/gcn_md$md.json').read().strip().splitlines()[-1]); print('maxd $md', d['value'], d.get('roofline',{}).get('frac'))
PY
done
