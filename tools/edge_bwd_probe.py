"""GPU box: time the EdgeConv max-pass backward (psg_edgeconv_bwd = edge pass + one small GEMM) on the graphs of the
28-block fixture, for one or more builds of the library:  python tools/edge_bwd_probe.py [lib.so ...]
Each library runs in a child process (ctypes binds one copy); prints the average time of the call with torch events."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(lib_path, rooms=4, reps=200):
    if lib_path.endswith("@atomic"):
        lib_path = lib_path[:-7]
        os.environ["PSG_GCN_EDGE_BWD"] = "atomic"
    import torch
    from pointsecguard_amd import _lib, runtime
    _lib.LIB_PATH = lib_path
    g = np.load(os.path.join(ROOT, "tests", "golden", "gcn28_room.npz"))["graphs"].astype(np.int32)
    N, GC = 4096, 64
    rng = np.random.default_rng(0)
    R = rooms * N
    nbr = torch.from_numpy(np.concatenate([g[5 + 5 * r] for r in range(rooms)], 0)).cuda()
    arg = torch.from_numpy((rng.integers(0, 16, (R, GC)) | np.where(rng.random((R, GC)) < 0.7, 0x80, 0)).astype(np.uint8)).cuda()
    dy = torch.randn(R, GC, device="cuda")
    sc = torch.rand(GC, device="cuda") + 0.5
    wt = torch.randn(GC, 2 * GC, device="cuda")
    dpq = torch.empty(R * 128, device="cuda")
    dx = torch.empty(R, GC, device="cuda")

    def call():
        _lib.call("psg_edgeconv_bwd", runtime.ptr(dy), GC, R, N, GC, runtime.ptr(nbr), runtime.ptr(arg), runtime.ptr(sc),
                  runtime.ptr(wt), runtime.ptr(dpq), runtime.ptr(dx), GC, runtime.stream())
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    lib = _lib.load()
    if hasattr(lib, "psg_debug_ebg_stamps"):
        st = np.zeros(256 * 16 * 8, np.uint64)
        lib.psg_debug_ebg_stamps(ctypes.c_void_p(st.ctypes.data), st.size)
        st = st.reshape(256, 16, 8).astype(np.int64)
        t0 = st[:, :, 0].min()
        d = st[:, :, :6] - t0
        names = ["start", "scan done", "after barrier", "lists ready", "first window staged", "walk done"]
        for k in range(6):
            print("   stamp %-20s median %7d  p10 %7d  p90 %7d  max %7d cycles after the first wave's start" % (
                names[k], np.median(d[:, :, k]), np.percentile(d[:, :, k], 10), np.percentile(d[:, :, k], 90), d[:, :, k].max()))
        hw = st[:, 0, 6]
        cu = [((int(h) >> 32) & 0xF, (int(h) >> 13) & 7, (int(h) >> 12) & 1, (int(h) >> 8) & 0xF) for h in hw]
        from collections import Counter
        cnt = Counter(cu)
        print("   workgroups: %d on %d distinct (xcc, se, sh, cu); most per CU: %d; per xcc: %s" % (
            len(cu), len(cnt), max(cnt.values()), sorted(Counter(c[0] for c in cu).items())))
        for x in range(8):
            sel = [i for i, c in enumerate(cu) if c[0] == x]
            if sel:
                t = st[sel][:, :, 0].min()
                print("   xcc %d: %d workgroups, start spread %d cycles, last walk-done %d cycles after the xcc's first start" % (
                    x, len(sel), st[sel][:, :, 0].max() - t, st[sel][:, :, 5].max() - t))
        print("   per-wave walk duration: median %d, max %d cycles; entries per wave: median %d max %d" % (
            np.median(st[:, :, 5] - st[:, :, 4]), (st[:, :, 5] - st[:, :, 4]).max(), np.median(st[:, :, 7]), st[:, :, 7].max()))
    print("%-40s %7.1f us per call (edge pass + 16384x128x64 GEMM), checksum %.6e" % (
        os.path.basename(lib_path) + ("@" + os.environ.get("PSG_GCN_EDGE_BWD", "")), e0.elapsed_time(e1) / reps * 1e3, float(dpq[:R * 2 * GC].double().sum())), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        one(sys.argv[2])
    else:
        libs = sys.argv[1:] or [os.path.join(ROOT, "pointsecguard_amd", "libpsg.so")]
        for lib in libs:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib if lib.startswith("/") else os.path.join(os.getcwd(), lib)], check=True)
