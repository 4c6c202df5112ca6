# GPU box: per-grid-size time of the RandLA-Net bench's 64x64-tile GEMM launches (which shapes the family's time sits in)
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
PSG_RLA_NO_GRAPH=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/rg -o rg -- python3 bench.py --allow-env-switches --workload randla --steps 8 --warmup 0 --concurrency 1 --randla-iters 6 --no-cpu-baseline --no-reference > $O/rla_gemm.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/rg/*kernel_trace.csv")[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gemm_rows_kernel<2, 2" in n and "1, 1>" in n:
        key = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]))
        a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("grid %4d x %2d (rows ~%6d, M ~%4d): %4d launches, %7.1f us each, %5.1f %% of the family" % (k[0], k[1], k[0] * 64, k[1] * 64, v[0], v[1] / v[0], 100 * v[1] / tot))
PY
rm -rf $O/rg
