#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_randla_net.py -x -q -m gpu > $O/rla_tests.log 2>&1; rc=$?
tail -4 $O/rla_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 400 python bench.py --workload randla --steps 24 --warmup 8 --no-cpu-baseline > $O/rla_l2.json 2> $O/rla_l2.err || { tail -5 $O/rla_l2.err; exit 1; }
python -c "
import json
d=json.loads([l for l in open('$O/rla_l2.json') if l.startswith('{')][-1])
print('randla', d['config']['distance_metric'], round(d['value'],2), d['unit'], 'l_inf:', d.get('l_inf'), 'unco:', d.get('uncoalesced_reference'))
print(d['roofline']['kernel'][:60], round(d['roofline']['frac'],3)); print(d['profiled_kernels_ms_per_iteration'])"
