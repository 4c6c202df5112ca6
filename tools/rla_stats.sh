# GPU box: rocprofv3 kernel-trace stats of the RandLA-Net bench at 8 clouds per launch -> top kernels
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs -o rs -- python3 bench.py --workload randla --steps 8 --warmup 0 --concurrency 1 --randla-iters ${ITERS:-20} --no-cpu-baseline --no-reference > $O/rla_stats.log 2>&1 || exit 1
cp $(find $O/rs -name '*kernel_stats.csv' | head -1) $O/rla_stats_kernel_stats.csv
rm -rf $O/rs
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/rla_stats_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:${ROWS:-30}]:
    print("%6.2f%% %6d calls %9.1f us  %s"%(float(r['Percentage']),int(r['Calls']),float(r['AverageNs'])/1e3,r['Name'][:100]))
print(tot/1e6, 'ms total')
PY
