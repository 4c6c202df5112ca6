# timing bisection of knn_select_kernel: builds variants that stop after stage 1 (load keys), 2 (sample bracket),
# 3 (bisection), 4 (compaction) and times them in the resgcn bench under rocprofv3 (results are wrong, timing only)
export TMPDIR=/tmp
for st in ${STAGES:-1 2 3 41 42 4 0}; do
  touch pointsecguard_amd/csrc/psg_resgcn.hip
  make -C pointsecguard_amd/csrc EXTRA=-DPSG_KNN_STAGE=$st > /dev/null 2>&1 || exit 1
  rm -rf gpurun_out/prof_ks
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ks -o k -- python3 bench.py --workload resgcn --steps 1 --warmup 1 --gcn-concurrency 1 > gpurun_out/prof_ks.log 2>&1 || exit 1
  python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/prof_ks/k_kernel_stats.csv')):
    if 'knn_select' in r['Name']: print('stage $st: avg %.1f us  min %.1f  max %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
"
done
rm -rf gpurun_out/prof_ks
