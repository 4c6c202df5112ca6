#!/bin/bash
# round 4: NU attacks with the device-side exit latch: parity tests, then the configs[2] bench in its three modes
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_nu.py tests/test_gpu_api.py -x -q -m gpu > $O/nu_tests.log 2>&1; rc=$?
tail -5 $O/nu_tests.log
[ $rc = 0 ] || exit $rc
for mode in per-room per-room-calls batch32; do
  timeout -k 10 300 python bench.py --workload tarnu --nu-mode $mode --steps 6 --warmup 2 --no-cpu-baseline > $O/tarnu_$mode.json 2> $O/tarnu_$mode.err || { tail -5 $O/tarnu_$mode.err; exit 1; }
  python -c "
import json
d=json.loads([l for l in open('$O/tarnu_$mode.json') if l.startswith('{')][-1])
print('$mode', round(d['value'],1), d['unit'], {k: d.get(k) for k in ('optimizer_steps_per_attack','attacks_reached_target','room_steps_per_sec')})"
done
