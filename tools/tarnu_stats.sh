# GPU box: rocprofv3 kernel-trace stats of one lockstep tar_NU step (32 rooms) -> top kernels
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ts -o ts -- python3 bench.py --workload tarnu --steps 2 --warmup 2 --nu-concurrency 1 --no-cpu-baseline --no-reference > $O/tarnu_stats.log 2>&1 || exit 1
cp $(find $O/ts -name '*kernel_stats.csv' | head -1) $O/tarnu_stats_kernel_stats.csv
rm -rf $O/ts
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/tarnu_stats_kernel_stats.csv")))
for r in rows[:${ROWS:-14}]:
    print("%6.2f%% %6d calls %9.1f us  %s"%(float(r['Percentage']),int(r['Calls']),float(r['AverageNs'])/1e3,r['Name'][:90]))
print(sum(float(r['TotalDurationNs']) for r in rows)/1e6, 'ms total')
PY
grep '^{' $O/tarnu_stats.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['ms_per_step'])"
