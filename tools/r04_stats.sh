#!/bin/bash
# kernel-trace stats of ONE bench workload (GPU box): tools/r04_stats.sh NAME bench-args...  ->  gpurun_out/r04/NAME_kernel_stats.csv
set -o pipefail
export TMPDIR=/tmp
name=$1; shift
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -o $name -- python3 bench.py "$@" > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }
grep '^{' $O/$name.log > $O/${name}_bench.json
cp $(find $O/tmp_$name -name '*kernel_stats.csv' | head -1) $O/${name}_kernel_stats.csv
rm -rf $O/tmp_$name
python3 - "$O/${name}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(__import__("os").environ.get("ROWS", "14"))]:
    print("%-78s calls %6s avg_us %9.1f pct %5s" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
