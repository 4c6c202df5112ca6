"""Where one NU optimiser step (32 rooms) spends its time: each call of the step sequence timed with a device
synchronise around it (so the sum overstates the pipelined step, but the split is what matters)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointsecguard_amd import _lib, runtime
from pointsecguard_amd.synthetic import make_rooms, rule_labels
from pointsecguard_amd.attacks.torchattacks.attacks.nu import ctypes_off

B, N = 32, 4096
sd = dict(np.load("tests/golden/pn2_weights.npz"))
model = runtime.PN2Model(runtime.fold_state_dict(sd))
ws = runtime.PN2Workspace(B, N, 11)
rooms = make_rooms(B, 1, structured=True)
x0 = torch.from_numpy(rooms).cuda()
labels = torch.from_numpy(rule_labels(rooms).astype(np.int32)).cuda()
starts = torch.zeros(11, 4, B, dtype=torch.int32, device="cuda")
ori = x0[:, :, 3:6].contiguous()
w = torch.zeros(B, N, 3, device="cuda"); m = torch.zeros_like(w); v = torch.zeros_like(w)
dlogp = torch.empty(B, N, 13, device="cuda"); dx0 = torch.empty(B, N, 9, device="cuda")
sgrad = torch.empty(N, 3, device="cuda"); pred = torch.empty(B, N, dtype=torch.int32, device="cuda")
scal = torch.zeros(3, device="cuda"); mask = torch.ones(N, dtype=torch.uint8, device="cuda")
st = runtime.stream
T = {}
def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return r
for rep in range(6):
    if rep == 1: T.clear()
    timed("plan_build(11 fwd)", lambda: ws.plan_build(x0, starts, 11))
    for step in range(10):
        timed("tanh_color", lambda: _lib.call("psg_nu_tanh_color", runtime.ptr(w), runtime.ptr(mask), B, N, runtime.ptr(x0), st()))
        logp = timed("forward", lambda: ws.forward(model, step, x0))
        timed("f_loss_grad", lambda: _lib.call("psg_nu_f_loss_grad", runtime.ptr(logp), runtime.ptr(labels), 0, B * N, 13, 0.0, 1.0, runtime.ptr(dlogp), runtime.ptr(scal[0:1]), runtime.ptr(pred), st()))
        timed("backward", lambda: ws.backward(model, step, dlogp, dx0))
        timed("smooth_knn", lambda: _lib.call("psg_smooth_knn", ctypes_off(x0, 3), 9, runtime.ptr(ori), 3, N, 5, runtime.ptr(scal[1:2]), runtime.ptr(sgrad), st()))
        timed("adam_step", lambda: _lib.call("psg_nu_adam_step", runtime.ptr(w), runtime.ptr(m), runtime.ptr(v), runtime.ptr(mask), runtime.ptr(dx0), runtime.ptr(x0), runtime.ptr(ori), runtime.ptr(sgrad), 1.0, 1.0, 0.01, 0.9, 0.999, 1e-8, step + 1, B, N, runtime.ptr(scal[2:3]), st()))
        timed("stats read-back", lambda: torch.stack([pred.eq(labels).sum().float(), scal[0], scal[1], scal[2]]).cpu())
n = 5 * 10
for k, t in T.items():
    print("%-22s %8.3f ms per step" % (k, t / n))
print("sum %.3f ms per step" % (sum(T.values()) / n))
