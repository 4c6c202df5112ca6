"""Attack API with the reference's import surface (`sys.path.append('attacks'); import torchattacks`,
PointNet/attacks/torchattacks/__init__.py:1-2)."""
from .attacks.nontarget import NB_attack, NU_attack
from .attacks.target import tar_NB_attack, tar_NU_attack

__all__ = ["NB_attack", "NU_attack", "tar_NB_attack", "tar_NU_attack"]
