"""CPU oracle for the whole-scene harness kernels (numpy).  TEST INFRASTRUCTURE ONLY (see oracle/pn2.py).

Restates, relative to /root/reference/PointNet/NB_nontarget_test_semseg.py:
  :55-62    add_vote        (the double loop, as one np.add.at)
  :184      torch.dist      (p = 2)
  :199-211  per-batch and :219-241 per-scene seen / correct / union counters and the mean IoU over classes present
Pinned by tests/golden/harness.npz (produced by the reference's own add_vote on the reference's own block slicing).
"""
import numpy as np


def add_vote(pool, point_idx, pred_label, weight):
    pool = np.array(pool, dtype=np.float64, copy=True)
    m = np.asarray(weight) != 0
    np.add.at(pool, (np.asarray(point_idx)[m].astype(np.int64), np.asarray(pred_label)[m].astype(np.int64)), 1.0)
    return pool


def counters(pred, gt, n_cls=13):
    pred, gt = np.asarray(pred).ravel(), np.asarray(gt).ravel()
    seen = np.array([np.sum(gt == l) for l in range(n_cls)])
    correct = np.array([np.sum((pred == l) & (gt == l)) for l in range(n_cls)])
    union = np.array([np.sum((pred == l) | (gt == l)) for l in range(n_cls)])
    return np.stack([seen, correct, union]).astype(np.int64)


def vote_stats(pool, labels, n_cls=13):
    pred = np.argmax(pool, 1)
    return counters(pred, labels, n_cls), pred


def miou(c):
    iou = c[1] / (c[2].astype(float) + 1e-6)
    return float(np.mean(iou[c[0] != 0]))


def l2_dist(a, b):
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return float(np.sqrt(np.sum(d * d)))
