"""Attack base class of the ResGCN harness (reference: ResGCN/sem_seg_dense/attacks/torchattacks/attack.py:4-176):
the same surface as the PointNet one, shared implementation."""
from pointsecguard_amd.attacks.torchattacks.attack import Attack  # noqa: F401
