#!/bin/bash
# GPU box: true kernel durations (rocprofv3 --kernel-trace --stats) of tools/edge_bwd_probe.py for each library given
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
for lib in "$@"; do
  n=$(basename ${lib%%@*} .so)$(echo $lib | grep -o '@.*' | tr '@' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$n -o $n -- python3 tools/edge_bwd_probe.py --child $PWD/$lib > $O/probe_$n.log 2>&1 || { tail -5 $O/probe_$n.log; exit 1; }
  f=$(find $O/tmp_$n -name '*kernel_stats.csv' | head -1)
  echo "== $lib"; python3 -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:5]: print('   %-70s calls %5s avg_us %8.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))"
  rm -rf $O/tmp_$n
done
