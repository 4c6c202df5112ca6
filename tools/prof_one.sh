#!/bin/bash
# kernel-trace stats of ONE bench workload (GPU box): tools/prof_one.sh NAME bench-args...  ->  gpurun_out/prof_NAME_kernel_stats.csv
set -o pipefail
export TMPDIR=/tmp
name=$1; shift
O=gpurun_out/prof_one
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o $name -- python3 bench.py "$@" > $O/$name.log 2>&1 || exit 1
grep '^{' $O/$name.log > gpurun_out/prof_${name}_bench.json
cp $(find $O/$name -name '*kernel_stats.csv' | head -1) gpurun_out/prof_${name}_kernel_stats.csv
rm -rf $O/$name
