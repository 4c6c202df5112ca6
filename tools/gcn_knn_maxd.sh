# GPU box: ResGCN-28 bench line with the bf16-prefilter kNN kernel serving dilations 1..MAXD (the exact kernel above)
mkdir -p gpurun_out/r03
for md in ${MAXDS:-0 2 3 4 5 6 8}; do
  PSG_GCN_KNN_BF_MAXD=$md timeout -k 10 200 python bench.py --allow-env-switches --workload resgcn --steps 12 --warmup 4 --no-cpu-baseline --no-reference > gpurun_out/r03/gcn_maxd$md.log 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("gpurun_out/r03/gcn_maxd$md.log") if x.startswith("{")][-1]
j=json.loads(l)
print("maxd $md", round(j["value"],2), "rooms/s", j.get("kernel_ms_per_iteration"))
PY
done
