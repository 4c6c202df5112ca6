"""Generate tests/golden/gcn_variants.npz by running the REFERENCE's DenseDeepGCN with its `block` / `conv` switches
(architecture.py:26-39, torch_vertex.py:44-49) -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gcn_variants.py

For every variant (4 blocks, one 1024-point room): neighbour tables of every graph convolution, the last block's
output, logits, mean cross-entropy and d cost / d input.  Weights are the seeded recipe
pointsecguard_amd.synthetic.gcn_state_dict(seed, 4, block, conv), loaded strict into the reference model (which pins
the key layout of every variant); the torch_cluster stand-in is the one make_golden_gcn.py documents.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/ResGCN"
sys.modules.setdefault("torch_cluster", types.ModuleType("torch_cluster"))
sys.modules["torch_cluster"].knn_graph = None
sys.path[:0] = [ROOT, REF, REF + "/sem_seg_dense"]
sys.dont_write_bytecode = True

from pointsecguard_amd.synthetic import gcn_state_dict, make_rooms, rule_labels  # noqa: E402

from architecture import DenseDeepGCN  # noqa: E402  (reference)

N_BLOCKS, NPT, SEED = 4, 1024, 31
VARIANTS = (("plain", "edge"), ("dense", "edge"), ("res", "mr"), ("plain", "mr"), ("dense", "mr"))


def main():
    torch.set_num_threads(4)
    r = make_rooms(1, 88)[:, :NPT].copy()
    y = rule_labels(r)
    x = torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 1))).unsqueeze(-1)
    out = {"room": r[0], "labels": y[0].astype(np.int16), "n_blocks": N_BLOCKS, "seed": SEED}
    for block, conv in VARIANTS:
        opt = SimpleNamespace(n_filters=64, k=16, act="relu", norm="batch", bias=True, epsilon=0.0, stochastic=True,
                              conv=conv, n_blocks=N_BLOCKS, block=block, in_channels=9, dropout=0.0, n_classes=13)
        m = DenseDeepGCN(opt)
        sd = gcn_state_dict(SEED, N_BLOCKS, block, conv)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m.eval()
        seen, last = {}, []

        def knn_hook(mod, inp, res):
            seen.setdefault(id(mod), res[0, 0].numpy().copy())
        hooks = [m.knn.register_forward_hook(knn_hook)]
        for blk in m.backbone:
            hooks.append(blk.body.dilated_knn_graph.register_forward_hook(knn_hook))
        hooks.append(m.backbone[-1].register_forward_hook(lambda mod, i, o: last.append(o.detach()[0, :, :, 0].T.contiguous().numpy())))
        xin = x.clone().requires_grad_(True)
        logits = m(xin)
        cost = torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(y))
        cost.backward()
        for h in hooks:
            h.remove()
        tag = "%s_%s_" % (block, conv)
        mods = [m.knn] + [blk.body.dilated_knn_graph for blk in m.backbone]
        for e, mod in enumerate(mods):
            out[tag + "nbr%d" % e] = seen[id(mod)].astype(np.int16)
        out[tag + "last"] = last[0][:, -64:]                     # the last block's 64 new channels
        out[tag + "logits"] = logits.detach()[0].T.contiguous().numpy()
        out[tag + "cost"] = np.float64(cost.item())
        out[tag + "dx"] = xin.grad[0, :, :, 0].T.contiguous().numpy()
        print(tag, "cost %.5f classes %d |dx|max %.3e" % (cost.item(), logits.argmax(1).unique().numel(),
                                                          xin.grad.abs().max().item()), flush=True)
    np.savez_compressed(os.path.join(HERE, "gcn_variants.npz"), **out)


if __name__ == "__main__":
    main()
