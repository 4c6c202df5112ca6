"""Generate tests/golden/randla_knn.npz with the REFERENCE's own k-NN code: oracle/_ref/libknn_ref.so is compiled by
`make -C oracle ref` from RandLA-Net/utils/nearest_neighbors/knn_.cxx where it lies (build container only).

    make -C oracle ref && python tests/golden/make_golden_randla.py

Cases: a structured synthetic room cloud (4096 points: planes, so near-ties occur), uniform noise, the k = 1 up-sampling
query of a sub-sampled cloud, and a cloud with duplicated points (exact ties).  Stored: inputs + the reference's indices.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import randla  # noqa: E402
from pointsecguard_amd.synthetic import make_rooms  # noqa: E402


def main():
    assert randla.have_ref(), "run `make -C oracle ref` first"
    out = {}
    room = make_rooms(2, 11, structured=True)[:, :, :3].copy()
    out["a_support"], out["a_k"] = room, 16
    out["a_idx"] = randla.knn_ref(room, room, 16)
    rng = np.random.default_rng(5)
    u = rng.random((1, 3000, 3), dtype=np.float32)
    qs = rng.random((1, 777, 3), dtype=np.float32)
    out["b_support"], out["b_query"], out["b_idx"] = u, qs, randla.knn_ref(u, qs, 5)
    sub = np.ascontiguousarray(room[:, :1024])
    out["c_idx"] = randla.knn_ref(sub, room, 1)              # up-sampling index: nearest sub-sampled point (k = 1)
    dup = u[:, :1500].copy()
    dup[:, 500:1000] = dup[:, :500]                          # 500 exact duplicates
    out["d_support"], out["d_idx"] = dup, randla.knn_ref(dup, dup, 16)
    np.savez_compressed(os.path.join(HERE, "randla_knn.npz"), **out)
    print({k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
