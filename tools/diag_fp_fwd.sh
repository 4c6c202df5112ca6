# needs a diagnostic library: make -C pointsecguard_amd/csrc clean && make -C pointsecguard_amd/csrc -j8 EXTRA=-DPSG_DIAG_BUILD
# (the default libpsg.so has no work-skipping switches; rebuild it plainly afterwards)
for d in 0 1 2 4 8 16 64; do PSG_DIAG=$d timeout -k 10 100 python bench.py --diag-build-ok --allow-env-switches --steps 4 --warmup 1 --no-cpu-baseline --concurrency 1 > gpurun_out/diag.log 2>&1 || exit 1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/diag.log') if l.startswith('{')][-1])
k=d['kernel_ms_per_attack']
print($d, 'fp1_head_fwd us', round(k['fp1_head_fwd']/40*1000,1), 'fp2_fwd', round(k['fp2_fwd']/40*1000,1), 'fp4_fwd', round(k['fp4_fwd']/40*1000,1))
"; done
