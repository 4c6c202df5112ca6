# GPU box: rocprofv3 kernel-trace stats of a short ResGCN-28 bench run -> gpurun_out/r03/gcn_stats_kernel_stats.csv (top rows printed)
export TMPDIR=/tmp
O=${O:-gpurun_out/r03}
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gs -o gs -- python3 bench.py --workload resgcn --steps 4 --warmup 4 --gcn-concurrency 1 --no-cpu-baseline --no-reference > $O/gcn_stats.log 2>&1 || exit 1
cp $(find $O/gs -name '*kernel_stats.csv' | head -1) $O/gcn_stats_kernel_stats.csv
rm -rf $O/gs
head -${ROWS:-9} $O/gcn_stats_kernel_stats.csv | cut -c1-120,121-400 | awk -F'",' '{print substr($1,1,70), $2}'
