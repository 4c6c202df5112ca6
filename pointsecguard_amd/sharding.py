"""Multi-GPU sharding of the attack workload: one process per GPU, rooms / scenes are independent
units (eval-mode BatchNorm, per-room operators and losses -- SURVEY.md section 8e), so the data path
has NO collective.  The only exchange is one all-reduce (RCCL over xGMI with backend "nccl"; gloo in
the CPU tests) of the int64 segmentation counters produced by psg_seg_stats.
"""
import numpy as np
import torch

NUM_CLASSES = 13


def shard_range(n_items, rank, world):
    """Contiguous split of n_items over `world` ranks (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_scenes(scene_ids, rank, world):
    """Round-robin scene assignment: whole scenes stay on one rank so the per-scene vote pool
    (NB_nontarget_test_semseg.py:140-141 of the reference) never crosses GPUs."""
    return [s for i, s in enumerate(scene_ids) if i % world == rank]


def draw_fps_starts_sharded(total_batch, n_point, n_forward, lo, hi):
    """FPS start draws for rooms [lo, hi) of a global batch: every rank draws the FULL batch vectors
    from an identically seeded CPU generator and keeps its columns, so the union over ranks equals the
    single-process stream (pointnet_util.py:75 draws torch.randint(0, N, (B,)) per level per forward)."""
    out = torch.empty(n_forward, 4, hi - lo, dtype=torch.int32)
    for f in range(n_forward):
        for lvl, n in enumerate((n_point, 1024, 256, 64)):
            out[f, lvl] = torch.randint(0, n, (total_batch,), dtype=torch.long)[lo:hi].to(torch.int32)
    return out


def fps_start_table_sharded(seed, n_forward, total_batch, lo, hi, n_point=4096):
    """The same for a whole attack's table of draws (bench.py: `iters` forwards planned at once): int32
    [n_forward][4][hi - lo] = columns lo..hi of the table a single process would draw for `total_batch` rooms from
    numpy's default_rng(seed), level by level.  Every rank passes the same seed and its own column range."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, n, (n_forward, total_batch))[:, lo:hi] for n in (n_point, 1024, 256, 64)],
                    axis=1).astype(np.int32)


def reduce_counters(counters, group=None):
    """Sum int64 counters over all ranks (no-op without an initialised process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters


def metrics_from_counters(counters):
    """counters int64 [3][n_cls] = (seen, intersection, union) -> dict with the reference's formulas:
    acc = sum(I)/sum(seen) (NB_nontarget_test_semseg.py:188-191), mIoU = mean over classes present of
    I/(U+1e-6) (:206-211), micro-IoU = sum(I)/sum(U) (ResGCN attacks.py:159-160)."""
    c = np.asarray(counters.cpu() if isinstance(counters, torch.Tensor) else counters, np.float64)
    seen, inter, union = c[0], c[1], c[2]
    iou = inter / (union + 1e-6)
    present = seen != 0
    return {"acc": float(inter.sum() / max(seen.sum(), 1.0)),
            "miou": float(iou[present].mean()) if present.any() else 0.0,
            "micro_iou": float(inter.sum() / max(union.sum(), 1.0)),
            "iou_per_class": iou.tolist()}


def seg_counters_host(pred, gt, n_cls=NUM_CLASSES):
    """Host-side (numpy) statement of the counter definition, for tests and tiny inputs."""
    pred, gt = np.asarray(pred).ravel(), np.asarray(gt).ravel()
    out = np.zeros((3, n_cls), np.int64)
    for l in range(n_cls):
        out[0, l] = np.sum(gt == l)
        out[1, l] = np.sum((pred == l) & (gt == l))
        out[2, l] = np.sum((pred == l) | (gt == l))
    return out
