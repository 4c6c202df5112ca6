#!/bin/bash
# rocprofv3 passes behind profiles/: kernel-trace stats of the bench command, then FETCH_SIZE / WRITE_SIZE
# in separate PMC passes (usage: tools/profile_round.sh TAG)
set -o pipefail
TAG=${1:-r01}
export TMPDIR=/tmp
O=gpurun_out/prof_$TAG
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pn2 -o pn2 -- python3 bench.py --steps 16 --warmup 8 --concurrency 1 --no-cpu-baseline --no-reference > $O/pn2.log 2>&1 || exit 1
grep '^{' $O/pn2.log > $O/pn2_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gcn -o gcn -- python3 bench.py --workload resgcn --steps 2 --warmup 1 --no-cpu-baseline > $O/gcn.log 2>&1 || exit 1
grep '^{' $O/gcn.log > $O/gcn_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/msg -o msg -- python3 bench.py --workload pointnet2_msg --steps 16 --warmup 8 --concurrency 1 > $O/msg.log 2>&1 || exit 1
grep '^{' $O/msg.log > $O/msg_bench.json
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 8 --warmup 8 --coalesce 8 --concurrency 1 --no-cpu-baseline --no-reference > $O/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py --steps 8 --warmup 8 --coalesce 8 --concurrency 1 --no-cpu-baseline --no-reference > $O/write.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json 64 > $O/pmc_traffic.txt
rm -rf $O/fetch $O/write   # raw per-dispatch rows are large; the summary is what is kept
find $O -name '*kernel_trace.csv' -delete
find $O -name '*agent_info.csv' -delete
ls -R $O | head -30
