"""CPU tests of the whole-scene harness: the host-side block slicing and the oracle of the device
kernels, both against tests/golden/harness.npz (outputs of the reference's ScannetDatasetWholeScene / add_vote)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness.npz")


@pytest.fixture(scope="module")
def g():
    return dict(np.load(GOLDEN))


def make_dataset(g):
    from pointsecguard_amd.harness import ScannetDatasetWholeScene
    names = [str(n) for n in g["file_list"]]
    scenes = {n: g["scene%d" % i].copy() for i, n in enumerate(names)}
    scenes["Area_1_office_9.npy"] = np.zeros((5, 7))      # a training-area file is ignored by split='test'
    scenes_before = {n: a.copy() for n, a in scenes.items()}
    ds = ScannetDatasetWholeScene(None, block_points=int(g["block_points"]), split="test", test_area=5, stride=0.5,
                                  block_size=1.0, padding=0.001, scenes=scenes)
    return ds, names, scenes, scenes_before


def test_block_slicing_matches_reference(g):
    ds, names, scenes, before = make_dataset(g)
    assert ds.file_list == names and len(ds) == len(names)
    # label weights are computed over the TEST scenes only; the reference's came from these same two files
    assert ds.labelweights.dtype == g["labelweights"].dtype
    assert np.array_equal(ds.labelweights, g["labelweights"])
    for si in range(len(names)):
        np.random.seed(100 + si)
        data_room, label_room, sample_weight, index_room = ds[si]
        for got, key in ((data_room, "data_room"), (label_room, "label_room"), (sample_weight, "sample_weight"),
                         (index_room, "index_room")):
            ref = g["%s%d" % (key, si)]
            assert got.shape == ref.shape and got.dtype == ref.dtype, key
            assert np.array_equal(got, ref), key
        assert np.array_equal(scenes[names[si]], before[names[si]]), "slicing must not modify the scene"


def test_file_backed_dataset(tmp_path, g):
    from pointsecguard_amd.harness import ScannetDatasetWholeScene
    names = [str(n) for n in g["file_list"]]
    for i, n in enumerate(names):
        np.save(tmp_path / n, g["scene%d" % i])
    np.save(tmp_path / "Area_2_x.npy", np.zeros((3, 7)))
    ds = ScannetDatasetWholeScene(str(tmp_path), block_points=int(g["block_points"]), split="test", test_area=5)
    assert sorted(ds.file_list) == sorted(names)
    si = ds.file_list.index(names[0])
    np.random.seed(100)
    assert np.array_equal(ds[si][0], g["data_room0"])
    assert len(ScannetDatasetWholeScene(str(tmp_path), split="train", test_area=5, block_points=64)) == 1


def test_oracle_votes_and_iou_match_reference(g):
    from oracle import harness as oh
    for si in range(len(g["file_list"])):
        pool = oh.add_vote(np.zeros_like(g["pool%d" % si]), g["index_room%d" % si], g["pred%d" % si], g["weight%d" % si])
        assert np.array_equal(pool, g["pool%d" % si])
        c, pred = oh.vote_stats(pool, g["scene%d" % si][:, 6])
        assert np.array_equal(pred, g["vote_pred%d" % si])
        assert np.array_equal(c[0], g["seen%d" % si]) and np.array_equal(c[1], g["correct%d" % si])
        assert np.array_equal(c[2], g["deno%d" % si])
        assert oh.miou(c) == pytest.approx(float(g["miou%d" % si]), abs=1e-12)
