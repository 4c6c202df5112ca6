"""FPS launch shapes (PSG_FPS_CFG) by problem count: microseconds per call of psg_fps for 4096 -> 1024 samples.
Usage: PSG_FPS_CFG=k python tools/fps_cfg_probe.py   (k = 0 auto, 1 = 1024 x 4, 2 = 512 x 8, 3 = 256 x 16)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pointsecguard_amd import runtime

rng = np.random.default_rng(0)
out = []
for P in (1, 8, 32, 64, 128, 320, 512, 640, 1280, 2560):
    xyz = torch.from_numpy(rng.random((P, 4096, 3), dtype=np.float32)).cuda()
    start = torch.from_numpy(rng.integers(0, 4096, P).astype(np.int32)).cuda()
    runtime.fps(xyz, 1024, start)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        runtime.fps(xyz, 1024, start)
    e1.record()
    torch.cuda.synchronize()
    out.append((P, e0.elapsed_time(e1) / 5 * 1e3))
print("PSG_FPS_CFG=%s" % os.environ.get("PSG_FPS_CFG", "0"), " ".join("P=%d:%.0fus" % o for o in out))
