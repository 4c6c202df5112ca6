"""Diagnostic (GPU box, tools/build_variant.sh tl psg_knn "-DPSG_KF_TL" + PSG_LIBRARY_OVERRIDE=build/libpsg_tl.so): real lengths of the fused kNN kernel's
workgroup-wide phases (mean cycles since kernel start at every workgroup barrier exit)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import _lib, runtime
lib = _lib.load()
rng = np.random.default_rng(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
f = rng.standard_normal((B, 4096, 64))
if os.environ.get("KNN_KIND", "plain") == "scaled":
    f = f * np.random.default_rng(3).uniform(0.2, 3.0, (1, 1, 64))
f = torch.from_numpy(f.astype(np.float32)).cuda()
ws = runtime.GCNWorkspace(B, 4096, 28)
buf = (ctypes.c_ulonglong * 32)()
# exact kernel: prune1 .. prune3+ are its cuts; prefilter kernel (round 4): slot 1 = sample stored, slot 2 = cut done
names = ("start", "prune1 / sample", "prune1 / cut done", "prune2", "prune2 done", "prune3+", "prune3+ done", "-", "stream end", "wave0 final end", "all final end", "slowest workgroup")
for d in [int(x) for x in os.environ.get("KNN_D", "1,4,9,17,27").split(",")]:
    ws.knn(f, d); lib.psg_dbg_knn_tl(buf)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        ws.knn(f, d)
    t1.record(); torch.cuda.synchronize()
    lib.psg_dbg_knn_tl(buf)
    if hasattr(lib, "psg_dbg_knn_ph"):
        ph = (ctypes.c_ulonglong * 32)()
        lib.psg_dbg_knn_ph(ph)
        pn = ("cut:load", "cut:minmax", "cut:hist", "cut:find", "cut:compact", "fin:load", "fin:params", "fin:hist", "fin:cum", "fin:search",
              "fin:collect", "fin:exact", "fin:rank")
        if any(ph[16 + i] for i in range(13)):      # (build with -DPSG_KF_TL -DPSG_KF_PH; the stamps inflate every phase by ~2k cycles)
            print("      phases (cycles per call of wave 0): " + "  ".join("%s %.0f" % (pn[i], ph[i] / ph[16 + i]) for i in range(13) if ph[16 + i]), flush=True)
    print("d=%2d  %.0f us per call  " % (d, t0.elapsed_time(t1) * 100) + "  ".join("%s %.0f" % (names[i], buf[i] / buf[16 + i]) for i in range(12) if buf[16 + i]), flush=True)
