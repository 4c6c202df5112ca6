"""GPU box: the kNN kernels on the REAL features of the fitted ResGCN-28 (one forward of 4 synthetic rooms): per block the
prefilter kernel's counters, equality with the exact kernel and time per call."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import runtime
from pointsecguard_amd.synthetic import gcn28_state_dict, make_rooms

B, N = 4, 4096
model = runtime.GCNModel(gcn28_state_dict(), 28)
rooms = make_rooms(B, 5000)
x0 = torch.from_numpy(np.ascontiguousarray(rooms)).cuda()          # [B, N, 9] point-major
os.environ["PSG_GCN_KNN_STATS"] = "1"
ws = runtime.GCNWorkspace(B, N, 28)
os.environ["PSG_GCN_KNN"] = "f32"
ex = runtime.GCNWorkspace(B, N, 28)
del os.environ["PSG_GCN_KNN"]
ws.forward(model, x0)
feats = ws.feats()                                                 # [B, N, 64 * 28]
for e in [int(v) for v in os.environ.get("BLOCKS", "1,2,3,5,9,14,20,27").split(",")]:
    x = feats[:, :, 64 * (e - 1):64 * e].contiguous()              # input of block e = output of block e - 1; dilation e
    sq = (x * x).sum(-1)
    ws.knn_stats()
    a = ws.knn(x, e)
    st = ws.knn_stats()
    b = ex.knn(x, e)
    bad = int((a != b).any(dim=-1).sum())
    t = []
    for w in (ws, ex):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            w.knn(x, e)
        t1.record(); torch.cuda.synchronize()
        t.append(t0.elapsed_time(t1) * 100)
    from pointsecguard_amd import _lib
    lib = _lib.load()
    if hasattr(lib, "psg_dbg_knn_tl"):
        import ctypes
        buf = (ctypes.c_ulonglong * 32)(); lib.psg_dbg_knn_tl(buf)
        ws.knn(x, e); lib.psg_dbg_knn_tl(buf)
        names = ("start", "sample", "cut done", "-", "-", "-", "-", "-", "stream end", "wave0 final end", "all final end", "slowest workgroup")
        print("      timeline: " + "  ".join("%s %.0f" % (names[i], buf[i] / buf[16 + i]) for i in range(12) if buf[16 + i]), flush=True)
        ph = (ctypes.c_ulonglong * 32)(); lib.psg_dbg_knn_ph(ph)
        pn = ("cut:load", "cut:minmax", "cut:hist", "cut:find", "cut:compact", "fin:load", "fin:params", "fin:hist", "fin:cum", "fin:search",
              "fin:collect", "fin:exact", "fin:rank")
        if any(ph[16 + i] for i in range(13)):
            print("      phases: " + "  ".join("%s %.0f" % (pn[i], ph[i] / ph[16 + i]) for i in range(13) if ph[16 + i]), flush=True)
    rows = max(st["rows"], 1)
    # spread of the 16 d-th nearest distances relative to the room's largest squared norm
    dist = torch.cdist(x[0, :256], x[0]) ** 2
    kth = dist.kthvalue(min(15 * e + 1, N), dim=1).values
    print("block %2d d=%2d |x|^2 max %.3g mean %.3g, KK-th distance / sqmax median %.4f | differing rows %d | bf16 %.0f us f32 %.0f us | "
          "tiles %d exact %d finalists/row %.1f cuts/row %.2f entries/row %.0f why %s"
          % (e, e, float(sq.max()), float(sq.mean()), float((kth / sq[0].max()).median()), bad, t[0], t[1], st["tiles"], st["exact_tiles"],
             st["finalists"] / rows, st["cuts"] / rows, st["entries"] / rows, {k: v for k, v in st.get("why", {}).items() if v}), flush=True)
