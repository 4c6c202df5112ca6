"""Helper of tests/test_gpu_harness.py::test_whole_scene_sharded_over_ranks: one rank of a gloo process group that
evaluates its share of the synthetic scenes on the (shared) GPU and writes the all-reduced totals."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import harness                                    # noqa: E402
from pointsecguard_amd.attacks import torchattacks                       # noqa: E402
from pointsecguard_amd.models.pointnet2_sem_seg import get_model         # noqa: E402


def synth_scene(seed, n, size_x, size_y):
    rng = np.random.default_rng(seed)
    xyz = rng.random((n, 3)) * np.array([size_x, size_y, 2.8])
    rgb = np.floor(rng.random((n, 3)) * 256.0)
    label = rng.integers(0, 13, n).astype(np.float64)
    return np.concatenate([xyz, rgb, label[:, None]], axis=1)


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    scenes = {"Area_5_a.npy": synth_scene(21, 3000, 1.4, 1.1), "Area_5_b.npy": synth_scene(22, 2500, 1.0, 1.3),
              "Area_5_c.npy": synth_scene(23, 2000, 1.2, 1.0)}
    ds = harness.ScannetDatasetWholeScene(None, block_points=1024, scenes=scenes)
    sd = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pn2_weights.npz")))
    net = get_model(13).cuda()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    # every scene is sliced and attacked from its own seed, so the result does not depend on which rank owns it
    results = None

    class SeededDataset:
        file_list = ds.file_list
        semantic_labels_list = ds.semantic_labels_list
        block_points = ds.block_points

        def __len__(self):
            return len(ds)

        def __getitem__(self, i):
            np.random.seed(100 + i)
            torch.manual_seed(100 + i)
            return ds[i]

    results = harness.evaluate_whole_scene(net, SeededDataset(), lambda m: torchattacks.NB_attack(m, eps=0.1, alpha=0.05, iters=2),
                                           batch_size=4, rank=rank, world=world, log=lambda *_: None)
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump({"counters": results["counters"].tolist(), "miou": results["miou"], "adv_miou": results["adv_miou"]}, fh)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
