"""pointsecguard_amd: MI355X-native implementation of PointSecGuard's data-parallel hot path.

Layout (mirrors the reference's import surface):
  models/pointnet_util.py, models/pointnet2_sem_seg.py   -- reference `PointNet/models/` API
  attacks/torchattacks/                                   -- reference `PointNet/attacks/torchattacks` API
  runtime.py, _lib.py                                     -- object layer + ctypes binding of libpsg.so
  csrc/                                                   -- HIP kernels and the C ABI (include/psg.h)
The HIP library is the only compute path; nothing falls back to CPU or eager PyTorch.
"""
__version__ = "0.1.0"
