import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
from pointsecguard_amd import runtime
from pointsecguard_amd.resgcn.sem_seg_dense.architecture import DenseDeepGCN
sd = dict(np.load("tests/golden/gcn_weights.npz")); g = dict(np.load("tests/golden/gcn_room.npz"))
opt = SimpleNamespace(n_filters=64, k=16, act="relu", norm="batch", bias=True, epsilon=0.0, stochastic=True, conv="edge", n_blocks=5, block="res", in_channels=9, dropout=0.0, n_classes=13)
net = DenseDeepGCN(opt); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net = net.cuda().eval()
inputs = torch.from_numpy(np.ascontiguousarray(g["room"].T[None, :, :, None])).cuda()
out = net(inputs)
a = out[0].T.cpu().numpy()
print("api maxdiff", np.abs(a - g["logits"]).max(), "close frac", (np.abs(a - g["logits"]).max(axis=1) <= 1e-3).mean())
m = runtime.GCNModel(sd, 5); ws = runtime.GCNWorkspace(1, 1024, 5)
lg = ws.forward(m, torch.from_numpy(g["room"][None]).cuda())[0].cpu().numpy()
print("direct maxdiff", np.abs(lg - g["logits"]).max(), "close frac", (np.abs(lg - g["logits"]).max(axis=1) <= 1e-3).mean())
t1 = runtime.gcn_tensor_list(sd, 5); t2 = runtime.gcn_tensor_list({k: v.detach().cpu() for k, v in net.state_dict().items()}, 5)
print([i for i, (x, y) in enumerate(zip(t1, t2)) if x.shape != y.shape or not np.array_equal(x, y)])
