"""Diagnostic (GPU box): the MSG network's colour gradient against the CPU oracle on the rooms of tests/test_gpu_msg.py::
test_msg_batch_vs_oracle, per room: sign agreement on the non-zero entries, the disagreeing entries' size relative to max |g|,
and how many entries differ by more than 1e-3 relative - for the library as built (PSG_PN2_SPLIT read from the environment)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pn2, pn2_msg
from pointsecguard_amd import _lib, runtime
from pointsecguard_amd.synthetic import make_rooms, rule_labels, msg_state_dict
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
room = dict(np.load(os.path.join(GOLD, "pn2msg_room.npz")))
sd = msg_state_dict(int(room["msg_seed"]))
model = runtime.PN2Model(runtime.fold_state_dict(sd, msg=True), arch=runtime.ARCH_MSG)
oracle = pn2_msg.PN2MsgOracle(sd)
dev = lambda a, dt=None: (torch.from_numpy(np.ascontiguousarray(a)).to(dt) if dt is not None else torch.from_numpy(np.ascontiguousarray(a))).cuda().contiguous()
for (B, N, seed) in ((3, 2048, 515), (2, 4096, 516)):
    rooms = make_rooms(B, seed, num_point=N)
    labels = rule_labels(rooms)
    torch.manual_seed(3)
    starts = np.stack([torch.randint(0, n, (B,)).numpy() for n in (N, 1024, 256, 64)]).astype(np.int32)
    ws = runtime.PN2Workspace(B, N, 1, arch=runtime.ARCH_MSG)
    x0 = dev(rooms)
    ws.plan_build(x0, dev(starts.reshape(1, 4, B), torch.int32), 1)
    logp = ws.forward(model, 0, x0)
    rows = B * N
    dlogp = torch.empty_like(logp)
    cost = torch.zeros(1, device="cuda")
    _lib.call("psg_ce_logp_grad", runtime.ptr(logp), runtime.ptr(dev(labels.astype(np.int32))), 0, rows, rows, 13, 1.0 / N, runtime.ptr(dlogp), runtime.ptr(cost), runtime.stream())
    dx0 = ws.backward(model, 0, dlogp)
    torch.cuda.synchronize()
    for b in range(B):
        geom = oracle.geometry(rooms[b, :, :3], starts[:, b])
        lp, cache = oracle.forward(rooms[b], geom)
        dl, _ = pn2.nll_logp_grad(lp, labels[b], 1.0 / N)
        ref = oracle.backward_color(cache, dl)
        ours = dx0[b, :, 3:6].cpu().numpy()
        nz = ref != 0
        agree = np.sign(ours[nz]) == np.sign(ref[nz])
        rel = np.abs(ours - ref)[nz] / np.abs(ref[nz])
        bad = np.abs(ref[nz][~agree]).max() / np.abs(ref).max() if not agree.all() else 0.0
        print("split=%s B=%d N=%d room %d: logp err %.2e, zero pattern equal %s, sign agreement %.5f (%d of %d differ, largest at %.1e max|g|), "
              "median rel %.1e, entries with rel > 1e-3: %d, > 1e-2: %d" % (os.environ.get("PSG_PN2_SPLIT", "1"), B, N, b, np.abs(logp[b].cpu().numpy() - lp).max(),
              np.array_equal(ours != 0, nz), agree.mean(), (~agree).sum(), nz.sum(), bad, np.median(rel), (rel > 1e-3).sum(), (rel > 1e-2).sum()), flush=True)
