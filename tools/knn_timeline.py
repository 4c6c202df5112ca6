"""Diagnostic (GPU box, make -C pointsecguard_amd/csrc EXTRA=-DPSG_KF_TL): real lengths of the fused kNN kernel's
workgroup-wide phases (mean cycles since kernel start at every workgroup barrier exit)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import _lib, runtime
lib = _lib.load()
rng = np.random.default_rng(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
f = torch.from_numpy(rng.standard_normal((B, 4096, 64)).astype(np.float32)).cuda()
ws = runtime.GCNWorkspace(B, 4096, 28)
buf = (ctypes.c_ulonglong * 32)()
names = ("start", "prune1", "prune1 done", "prune2", "prune2 done", "prune3+", "prune3+ done", "-", "stream end", "wave0 final end", "all final end")
for d in [int(x) for x in os.environ.get("KNN_D", "1,4,9,17,27").split(",")]:
    ws.knn(f, d); lib.psg_dbg_knn_tl(buf)
    for _ in range(10):
        ws.knn(f, d)
    lib.psg_dbg_knn_tl(buf)
    print("d=%2d  " % d + "  ".join("%s %.0f" % (names[i], buf[i] / buf[16 + i]) for i in range(11) if buf[16 + i]), flush=True)
