"""Diagnostic (GPU box): where the fused kNN kernel spends its cycles, per buffer class.  Needs the stamp build:
    make -C pointsecguard_amd/csrc clean && make -C pointsecguard_amd/csrc EXTRA=-DPSG_KF_STAMP
Shares, not lengths: the stamps' waits forbid overlaps the real kernel has (cdna_hip_programming.md section 7)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import _lib, runtime
lib = _lib.load()
rng = np.random.default_rng(1)
f = torch.from_numpy(rng.standard_normal((1, 4096, 64)).astype(np.float32)).cuda()
ws = runtime.GCNWorkspace(1, 4096, 28)
buf = (ctypes.c_ulonglong * 16)()
names = ("stream", "barrier wait", "appends", "prune", "final", "prune events", "roll-backs")
for d in (1, 4, 9, 17, 27):
    ws.knn(f, d); lib.psg_dbg_knn_stamps(buf)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        ws.knn(f, d)
    t1.record(); torch.cuda.synchronize()
    lib.psg_dbg_knn_stamps(buf)
    v = [buf[i] / 2560.0 for i in range(7)]     # per workgroup (256 workgroups x 10 launches)
    print("d=%2d  %.1f us/launch (incl. prep) | cycles per workgroup: " % (d, t0.elapsed_time(t1) * 100) +
          ", ".join("%s %.0f" % (n, x) for n, x in zip(names[:5], v[:5])) + " | events %.2f roll-backs %.2f" % (v[5], v[6]) +
          " | stream split: operands %.0f, mfma %.0f, keys %.0f, alloc %.0f" % (buf[8] / 2560.0, buf[9] / 2560.0, buf[10] / 2560.0,
                                                                               v[0] - (buf[8] + buf[9] + buf[10]) / 2560.0) + " | whole kernel %.0f" % (buf[13] / 2560.0), flush=True)
