# needs a diagnostic library: make -C pointsecguard_amd/csrc clean && make -C pointsecguard_amd/csrc -j8 EXTRA=-DPSG_DIAG_BUILD
# (the default libpsg.so has no work-skipping switches; rebuild it plainly afterwards)
# timing bisection of the SA/FP module kernels: PSG_DIAG bits 1 = no gather prologue, 8 = no layers, 16 = no barriers,
# 32 = no pooled/scatter epilogue, 64 = k8 = 4 in every layer (FP only)
for d in 0 1 8 16 32 41; do PSG_DIAG=$d timeout -k 10 100 python bench.py --diag-build-ok --allow-env-switches --steps 4 --warmup 1 --no-cpu-baseline --no-reference --no-secondary --concurrency 1 > gpurun_out/diag.log 2>&1 || exit 1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/diag.log') if l.startswith('{')][-1])
k=d['kernel_ms_per_attack']
print($d, ' '.join('%s %.1f' % (n, k[n]/40*1000) for n in ('sa1_fwd','sa2_fwd','sa3_fwd','sa4_fwd','sa1_bwd','sa2_bwd','sa3_bwd','sa4_bwd','fp2_bwd','fp1_head_bwd')))
"; done
