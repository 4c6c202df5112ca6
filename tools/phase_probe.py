"""Per-wave phase stamps of one feature-propagation forward kernel (PSG_DIAG = 512 | (level+1) << 16):
cycles spent in each layer's compute and at each barrier, averaged over waves.  usage: phase_probe.py LEVEL"""
import ctypes, os, sys
import numpy as np
LVL = int(sys.argv[1]) if len(sys.argv) > 1 else 0
os.environ["PSG_DIAG"] = str(512 | ((LVL + 1) << 16))
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
logp = None
for _ in range(30):
    logp = ws.forward(model, 0, x0)
    dl = torch.zeros_like(logp)
    ws.backward(model, 0, dl)
torch.cuda.synchronize()
nwg = {0: 1024, 1: 256, 2: 64, 3: 16}[LVL]
nl = {0: 5, 1: 2, 2: 2, 3: 2}[LVL]
n = nwg * 8 * 16
buf = (ctypes.c_ulonglong * n)()
_lib.call("psg_pn2_debug_read", ws.handle, buf, n)
a = np.array(buf[:], dtype=np.uint64).reshape(-1, 16).astype(np.float64)
a = a[a[:, 0] > 0]
print("waves", len(a))
print("prologue            mean %8.0f cycles" % (a[:, 0] - a[:, 15]).mean())
t = a[:, :1 + 2 * nl]
d = np.diff(t, axis=1)
names = sum([["L%d compute" % l, "L%d barrier" % l] for l in range(nl)], [])
for i, nm in enumerate(names):
    print("%-18s mean %8.0f cycles  p10 %8.0f p90 %8.0f" % (nm, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("total from entry   ", (a[:, 2 * nl] - a[:, 15]).mean())
