"""GPU box: time psg_ball_query per SA level at the bench's plan size (2560 problems; level 0 shares 64 clouds), grid kernel
(default) or full scan (PSG_BALL_QUERY=scan): python tools/ball_query_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import _lib, runtime
from pointsecguard_amd.synthetic import make_rooms
P, B = 2560, 64
rooms = make_rooms(B, 5)
xyz0 = torch.from_numpy(np.ascontiguousarray(rooms[:, :, :3])).cuda()
rng = np.random.default_rng(0)
ctx = runtime.context(xyz0.device)
levels = ((4096, 1024, 0.1), (1024, 256, 0.2), (256, 64, 0.4), (64, 16, 0.8))
cur, ncl = xyz0, B
for n, s, r in levels:
    idx = torch.from_numpy(np.stack([rng.choice(n, s, replace=False) for _ in range(P)]).astype(np.int64)).cuda()
    src = cur if ncl == P else cur.repeat(P // ncl, 1, 1)
    new = torch.gather(src, 1, idx[:, :, None].expand(-1, -1, 3)).contiguous()
    out = torch.empty(P, s, 32, dtype=torch.int32, device="cuda")
    r2 = float(np.float32(r * r))
    def call():
        _lib.call("psg_ball_query", ctx, runtime.ptr(cur), ncl, runtime.ptr(new), P, n, s, r2, 32, runtime.ptr(out), runtime.stream())
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): call()
    e1.record(); torch.cuda.synchronize()
    print("%s N=%4d S=%4d r=%.1f: %.3f ms per launch of %d problems" % (os.environ.get("PSG_BALL_QUERY", "grid"), n, s, r, e0.elapsed_time(e1) / 5, P), flush=True)
    cur, ncl = new, P
