"""In-kernel shader clock under load: PSG_DIAG=256 makes the fp1+head forward kernel stamp
{s_memtime, s_memrealtime} per workgroup; clock = d(memtime)/d(memrealtime) x 100 MHz.
usage: clock_probe.py [device batch rooms]  (runs whole 40-iteration attacks so the clock is the sustained one)"""
import ctypes, os, sys
import numpy as np
os.environ["PSG_DIAG"] = "256"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointsecguard_amd import _lib, runtime
from pointsecguard_amd.synthetic import make_rooms

sd = dict(np.load("tests/golden/pn2_weights.npz"))
model = runtime.PN2Model(runtime.fold_state_dict(sd))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ITERS = 40
ws = runtime.PN2Workspace(B, 4096, ITERS)
rooms = make_rooms(B, 1)
images = torch.from_numpy(np.ascontiguousarray(rooms.transpose(0, 2, 1))).cuda()
labels = torch.zeros(B, 4096, dtype=torch.int32, device="cuda")
starts = torch.zeros(ITERS, 4, B, dtype=torch.int32, device="cuda")
adv = torch.empty_like(images)
for _ in range(6):          # sustained load: six full 40-iteration attacks back to back, stamps of the last launch
    ws.nb_attack(model, images, labels, starts, 0.05, 2 / 255, ITERS, out=adv)
torch.cuda.synchronize()
n = 128 * B * 4
buf = (ctypes.c_ulonglong * n)()
_lib.call("psg_pn2_debug_read", ws.handle, buf, n)
a = np.array(buf[:], dtype=np.uint64).reshape(-1, 4).astype(np.float64)
dt, dr = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
ok = dr > 0
print("workgroups", ok.sum(), "median kernel-resident time us", np.median(dr[ok]) / 100.0,
      "median clock GHz", np.median(dt[ok] / dr[ok]) * 0.1, "p10/p90", np.percentile(dt[ok] / dr[ok], [10, 90]) * 0.1)
r0, r1 = a[ok, 1], a[ok, 3]
print("span us (first start -> last end)", (r1.max() - r0.min()) / 100.0, "start spread us", (r0.max() - r0.min()) / 100.0)
starts_sorted = np.sort(r0 - r0.min()) / 100.0
print("start time percentiles us", np.percentile(starts_sorted, [10, 25, 50, 75, 90, 100]))
