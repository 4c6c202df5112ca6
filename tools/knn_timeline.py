"""Diagnostic (GPU box, make -C pointsecguard_amd/csrc EXTRA=-DPSG_KF_TL): real lengths of the fused kNN kernel's
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
names = ("start", "prune1", "prune1 done", "prune2", "prune2 done", "prune3+", "prune3+ done", "-", "stream end", "wave0 final end", "all final end", "slowest workgroup")
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
        pn = ("cut:load", "cut:k2", "cut:hist", "cut:scan", "cut:compact", "fin:load", "fin:k2", "fin:hist", "fin:cum", "fin:search",
              "fin:collect", "fin:exact", "fin:rank")
        print("      phases (cycles per call of wave 0): " + "  ".join("%s %.0f" % (pn[i], ph[i] / ph[16 + i]) for i in range(13) if ph[16 + i]), flush=True)
    print("d=%2d  %.0f us per call  " % (d, t0.elapsed_time(t1) * 100) + "  ".join("%s %.0f" % (names[i], buf[i] / buf[16 + i]) for i in range(12) if buf[16 + i]), flush=True)
