#!/bin/bash
# round 5, call H: do the replayed hipGraphs of the ResGCN / RandLA-Net / NU loops still pay at the bench's launch shapes?
mkdir -p gpurun_out/r5n
run() { # name env workload-args
  local name=$1 envs=$2; shift 2
  env $envs python bench.py --allow-env-switches --no-cpu-baseline --no-reference "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],2), d['config'].get('hipgraph'))"
}
run gcn_graph "X=1" --workload resgcn --steps 24 --warmup 8
run gcn_eager "PSG_GCN_NO_GRAPH=1" --workload resgcn --steps 24 --warmup 8
run rla_graph "X=1" --workload randla
run rla_eager "PSG_RLA_NO_GRAPH=1" --workload randla
run nu_graph "X=1" --workload tarnu --steps 16 --warmup 8
run nu_eager "PSG_NU_NO_GRAPH=1" --workload tarnu --steps 16 --warmup 8
