"""Per-wave phase stamps of the fp1+head forward kernel (PSG_DIAG=512): cycles spent in each layer's
compute and at each barrier, averaged over waves."""
import ctypes, os, sys
import numpy as np
os.environ["PSG_DIAG"] = "512"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointsecguard_amd import _lib, runtime
from pointsecguard_amd.synthetic import make_rooms

sd = dict(np.load("tests/golden/pn2_weights.npz"))
model = runtime.PN2Model(runtime.fold_state_dict(sd))
B = 8
ws = runtime.PN2Workspace(B, 4096, 1)
x0 = torch.from_numpy(make_rooms(B, 1)).cuda()
starts = torch.zeros(1, 4, B, dtype=torch.int32, device="cuda")
ws.plan_build(x0, starts, 1)
for _ in range(50):
    ws.forward(model, 0, x0)
torch.cuda.synchronize()
n = 1024 * 4 * 16
buf = (ctypes.c_ulonglong * n)()
_lib.call("psg_pn2_debug_read", ws.handle, buf, n)
a = np.array(buf[:], dtype=np.uint64).reshape(-1, 16).astype(np.float64)
a = a[a[:, 0] > 0]
t = a[:, :11] - a[:, :1]
names = ["start"] + sum([["L%d done" % l, "L%d barrier" % l] for l in range(5)], [])
d = np.diff(t, axis=1)
for i in range(10):
    print("%-12s mean %8.0f cycles  p10 %8.0f p90 %8.0f" % (names[i + 1], d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("total", t[:, 10].mean())
