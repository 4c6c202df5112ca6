#!/bin/bash
# rocprofv3 passes behind profiles/<TAG>_*: kernel-trace stats of every bench workload, then FETCH_SIZE / WRITE_SIZE and the
# matrix-busy counters in SEPARATE PMC passes (program directly after `--`, no trace domains with --pmc).
# usage (on the GPU box, from the repo root): tools/profile_round.sh TAG   ->  gpurun_out/prof_TAG/, copy what is to be kept
set -o pipefail
TAG=${1:-r05}
WHAT=${2:-all}     # all | stats | stats_pn2 | pmc_pn2 | pmc_gcn | pmc_knn | pmc_rla | pmc_tarnu | pmc_msg
export TMPDIR=/tmp
O=gpurun_out/prof_$TAG
mkdir -p $O
stats() {   # name, bench args...
    local name=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o $name -- python3 bench.py --allow-env-switches --no-process-group "$@" > $O/$name.log 2>&1 || return 1
    grep '^{' $O/$name.log > $O/${name}_bench.json
    cp $(find $O/$name -name '*kernel_stats.csv' | head -1) $O/${name}_kernel_stats.csv
    rm -rf $O/$name
}
if [ $WHAT = all ] || [ $WHAT = stats ] || [ $WHAT = stats_pn2 ]; then
stats pn2 --steps 16 --warmup 8 --concurrency 1 --no-cpu-baseline --no-reference --no-secondary || exit 1
fi
if [ $WHAT = all ] || [ $WHAT = stats ]; then
stats gcn --workload resgcn --steps 4 --warmup 4 --gcn-concurrency 1 --no-cpu-baseline --no-reference || exit 1
stats msg --workload pointnet2_msg --steps 16 --warmup 8 --concurrency 1 --no-cpu-baseline || exit 1
stats tarnu --workload tarnu --steps 2 --warmup 1 --nu-concurrency 1 --no-cpu-baseline || exit 1
stats randla --workload randla --steps 12 --warmup 12 --rla-coalesce 12 --rla-concurrency 1 --no-cpu-baseline --no-reference || exit 1
fi
pmc() {     # name, counters (quoted), bench args...
    local name=$1 ctr=$2; shift 2
    # (the attack loops stay eager in PMC passes: per-dispatch counter rows need per-dispatch launches)
    # (--no-process-group: the one-rank RCCL communicator of round 5 stays out of counter collection)
    PSG_GCN_NO_GRAPH=1 PSG_RLA_NO_GRAPH=1 PSG_NU_NO_GRAPH=1 timeout -k 10 150 rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o p -- python3 bench.py --allow-env-switches --no-process-group "$@" > $O/$name.log 2>&1 || return 1
}
PN2="--steps 8 --warmup 8 --coalesce 8 --concurrency 1 --no-cpu-baseline --no-reference --no-secondary"
# (a counter pass serialises every dispatch: one 4-room launch of the ResGCN attack = 15k dispatches is plenty)
GCN="--workload resgcn --steps 4 --warmup 0 --gcn-concurrency 1 --no-cpu-baseline --no-reference"
if [ $WHAT = all ] || [ $WHAT = pmc_pn2 ]; then
pmc fetch FETCH_SIZE $PN2 || exit 1
pmc write WRITE_SIZE $PN2 || exit 1
python3 tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json 64 > $O/pmc_traffic.txt
pmc mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" $PN2 || exit 1
python3 tools/pmc_mfma.py $O/mfma $O/pmc_mfma.json > $O/pmc_mfma.txt
fi
if [ $WHAT = all ] || [ $WHAT = pmc_gcn ]; then
pmc gfetch FETCH_SIZE $GCN || exit 1
pmc gwrite WRITE_SIZE $GCN || exit 1
python3 tools/pmc_traffic.py $O/gfetch $O/gwrite $O/pmc_traffic_gcn.json 4 > $O/pmc_traffic_gcn.txt
# (round 2 saw this pass stop twice at the end of a long series of profiler sessions; on a fresh box it finishes, traced
# launch by launch and plain, with either kNN block order: tools/gmfma_trace.sh, profiles/r03_gmfma_trace_*)
pmc gmfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" $GCN || exit 1
python3 tools/pmc_mfma.py $O/gmfma $O/pmc_mfma_gcn.json > $O/pmc_mfma_gcn.txt
fi
if [ $WHAT = all ] || [ $WHAT = pmc_rla ]; then
RLA="--workload randla --steps 12 --warmup 0 --rla-coalesce 12 --rla-concurrency 1 --randla-iters 6 --no-cpu-baseline --no-reference"
pmc rfetch FETCH_SIZE $RLA || exit 1
pmc rwrite WRITE_SIZE $RLA || exit 1
python3 tools/pmc_traffic.py $O/rfetch $O/rwrite $O/pmc_traffic_randla.json 12 > $O/pmc_traffic_randla.txt
fi
if [ $WHAT = all ] || [ $WHAT = pmc_tarnu ]; then
# configs[2] in lockstep: two 32-room steps per call = 64 rooms per launch of the network kernels, one call in flight, eager windows
TNU="--workload tarnu --steps 2 --warmup 0 --nu-concurrency 1 --nu-steps 12 --no-cpu-baseline --no-reference"
pmc tfetch FETCH_SIZE $TNU || exit 1
pmc twrite WRITE_SIZE $TNU || exit 1
python3 tools/pmc_traffic.py $O/tfetch $O/twrite $O/pmc_traffic_tarnu.json 64 > $O/pmc_traffic_tarnu.txt
fi
if [ $WHAT = all ] || [ $WHAT = pmc_msg ]; then
MSG="--workload pointnet2_msg --steps 8 --warmup 8 --coalesce 8 --concurrency 1 --no-cpu-baseline --no-reference"
pmc mfetch FETCH_SIZE $MSG || exit 1
pmc mwrite WRITE_SIZE $MSG || exit 1
python3 tools/pmc_traffic.py $O/mfetch $O/mwrite $O/pmc_traffic_msg.json 64 > $O/pmc_traffic_msg.txt
fi
if [ $WHAT = all ] || [ $WHAT = pmc_knn ]; then
# the matrix-busy counters of the fused kNN kernel on its stand-alone launches (4 rooms per launch, d = 1, 4, 9, 17, 27)
timeout -k 10 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/kmfma -o p -- python3 tools/knn_time.py 4 > $O/kmfma.log 2>&1 || exit 1
python3 tools/pmc_mfma.py $O/kmfma $O/pmc_mfma_knn.json > $O/pmc_mfma_knn.txt
fi
rm -rf $O/fetch $O/write $O/gfetch $O/gwrite $O/mfma $O/gmfma $O/kmfma $O/rfetch $O/rwrite $O/tfetch $O/twrite $O/mfetch $O/mwrite   # raw per-dispatch rows are large; the summaries are what is kept
ls $O
