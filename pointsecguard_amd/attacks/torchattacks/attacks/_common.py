"""Shared plumbing of the colour-perturbation attacks: label / mask normalisation and access to the
libpsg workspace of a get_model instance."""
import numpy as np
import torch


def psg_model(model):
    if not hasattr(model, "_packed") or not hasattr(model, "_workspace"):
        raise TypeError("pointsecguard_amd attacks drive the gfx950 kernels of pointsecguard_amd.models."
                        "pointnet2_sem_seg.get_model; got %s" % type(model).__name__)
    return model


def labels_to_device(labels, device, pin=False):
    """The harness hands labels over as float64 numpy (NB_nontarget_test_semseg.py:171); the reference
    casts with torch.tensor(labels, dtype=int64) (nontarget.py:25)."""
    if isinstance(labels, torch.Tensor):
        t = labels.detach()
    else:
        t = torch.from_numpy(np.ascontiguousarray(labels))
    from pointsecguard_amd.models.pointnet2_sem_seg import upload
    return upload(t.to(torch.int32).contiguous(), device, pin=pin)


def mask_to_device(mask, n_point, device):
    if isinstance(mask, torch.Tensor):
        m = mask.detach().to(torch.bool).cpu().numpy()
    else:
        m = np.asarray(mask).astype(bool)
    if m.shape != (n_point,):
        raise ValueError("mask must be a boolean vector of length %d, got shape %s" % (n_point, m.shape))
    from pointsecguard_amd.models.pointnet2_sem_seg import upload
    return upload(torch.from_numpy(m.astype(np.uint8)), device)
