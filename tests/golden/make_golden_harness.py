"""Golden vectors for the whole-scene evaluation harness, produced by the REFERENCE code.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_harness.py

Runs, on a small synthetic S3DIS-format scene written to a scratch directory inside this repository:
  * PointNet/data_utils/S3DISDataLoader.py: ScannetDatasetWholeScene.__init__/__getitem__  (block slicing, label weights)
  * PointNet/NB_nontarget_test_semseg.py:55-62  add_vote
  * the per-scene IoU arithmetic of NB_nontarget_test_semseg.py:219-241 (restated inline below from the reference's
    own expressions on the reference's own outputs; the script keeps it inside main()).
Only data (inputs and outputs) is stored in tests/golden/harness.npz.  Nothing is written under /root/reference.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/PointNet"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synth_scene(seed, n, size_x, size_y):
    rng = np.random.default_rng(seed)
    xyz = rng.random((n, 3)) * np.array([size_x, size_y, 2.8])
    rgb = np.floor(rng.random((n, 3)) * 256.0)
    label = rng.integers(0, 13, n).astype(np.float64)
    label[rng.random(n) < 0.3] = 2.0          # an over-represented class, so the label weights are not flat
    return np.concatenate([xyz, rgb, label[:, None]], axis=1)


def main():
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "attacks"))   # the script appends a cwd-relative path for `import torchattacks`
    from data_utils.S3DISDataLoader import ScannetDatasetWholeScene
    import importlib
    cwd = os.getcwd()
    script = importlib.import_module("NB_nontarget_test_semseg")
    os.chdir(cwd)
    scratch = tempfile.mkdtemp(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT)
    try:
        scenes = {"Area_5_office_1.npy": synth_scene(1, 2600, 2.0, 1.5), "Area_5_hallway_2.npy": synth_scene(2, 1500, 1.2, 2.3),
                  "Area_1_office_9.npy": synth_scene(3, 300, 1.0, 1.0)}
        for name, arr in scenes.items():
            np.save(os.path.join(scratch, name), arr)
        block_points = 256
        ds = ScannetDatasetWholeScene(scratch + os.sep, block_points=block_points, split="test", test_area=5, stride=0.5,
                                      block_size=1.0, padding=0.001)
        out = {"block_points": np.int64(block_points), "file_list": np.array(ds.file_list), "labelweights": ds.labelweights}
        for si, fname in enumerate(ds.file_list):
            out["scene%d" % si] = scenes[fname]
            np.random.seed(100 + si)
            data_room, label_room, sample_weight, index_room = ds[si]
            out["data_room%d" % si] = data_room
            out["label_room%d" % si] = label_room
            out["sample_weight%d" % si] = sample_weight
            out["index_room%d" % si] = index_room
            # votes: random predictions, a few zero weights, through the reference's add_vote
            rng = np.random.default_rng(7 + si)
            pred = rng.integers(0, 13, label_room.shape)
            weight = sample_weight.copy()
            weight[rng.random(weight.shape) < 0.1] = 0.0
            n_pts = scenes[fname].shape[0]
            pool = script.add_vote(np.zeros((n_pts, 13)), index_room, pred, weight)
            out["pred%d" % si] = pred
            out["weight%d" % si] = weight
            out["pool%d" % si] = pool
            # NB_nontarget_test_semseg.py:219-241 on these votes
            whole_scene_label = ds.semantic_labels_list[si]
            pred_label = np.argmax(pool, 1)
            seen = np.array([np.sum(whole_scene_label == l) for l in range(13)])
            correct = np.array([np.sum((pred_label == l) & (whole_scene_label == l)) for l in range(13)])
            deno = np.array([np.sum((pred_label == l) | (whole_scene_label == l)) for l in range(13)])
            iou_map = correct / (deno.astype(float) + 1e-6)
            out["vote_pred%d" % si] = pred_label
            out["seen%d" % si] = seen
            out["correct%d" % si] = correct
            out["deno%d" % si] = deno
            out["miou%d" % si] = np.float64(np.mean(iou_map[seen != 0]))
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "harness.npz"), **out)
        print("wrote harness.npz:", {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("data_room")})
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
