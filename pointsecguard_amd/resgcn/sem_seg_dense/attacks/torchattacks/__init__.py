"""ResGCN attack API (reference: ResGCN/sem_seg_dense/attacks/torchattacks/__init__.py:1-2)."""
from .attacks.colper import NB_attack, NU_attack
from .attacks.tcolper import tar_NB_attack, tar_NU_attack

__all__ = ["NB_attack", "NU_attack", "tar_NB_attack", "tar_NU_attack"]
