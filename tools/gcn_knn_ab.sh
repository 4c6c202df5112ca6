# GPU box: ResGCN-28 bench line under each kNN kernel choice (default = prefilter for d <= 3, exact above)
mkdir -p gpurun_out/r03
for cfg in "f32:PSG_GCN_KNN=f32" "default:PSG_X=0" "bf16:PSG_GCN_KNN=bf16"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout -k 10 200 python bench.py --allow-env-switches --workload resgcn --steps 12 --warmup 4 --no-cpu-baseline --no-reference > gpurun_out/r03/gcn_$name.log 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("gpurun_out/r03/gcn_$name.log") if x.startswith("{")][-1]
j=json.loads(l)
print("$name", round(j["value"],2), "rooms/s", j.get("kernel_ms_per_iteration"))
PY
done
