"""CPU tests of the multi-GPU path: world_size-2 gloo run of the shard + counter all-reduce logic."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pointsecguard_amd import sharding


def test_shard_range_covers_everything():
    for n in (1, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_shard_scenes_round_robin():
    ids = list(range(68))
    got = sorted(sum((sharding.shard_scenes(ids, r, 8) for r in range(8)), []))
    assert got == ids


def test_sharded_fps_draws_equal_single_process():
    torch.manual_seed(9)
    full = sharding.draw_fps_starts_sharded(32, 4096, 3, 0, 32)
    parts = []
    for r in range(8):
        torch.manual_seed(9)
        lo, hi = sharding.shard_range(32, r, 8)
        parts.append(sharding.draw_fps_starts_sharded(32, 4096, 3, lo, hi))
    assert torch.equal(torch.cat(parts, dim=2), full)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    gt = rng.integers(0, 13, (32, 4096))
    pred = np.where(rng.random((32, 4096)) < 0.6, gt, rng.integers(0, 13, (32, 4096)))
    lo, hi = sharding.shard_range(32, rank, world)
    local = torch.from_numpy(sharding.seg_counters_host(pred[lo:hi], gt[lo:hi]))
    total = sharding.reduce_counters(local.clone())
    if rank == 0:
        q.put((total.numpy(), sharding.seg_counters_host(pred, gt)))
    dist.barrier()
    dist.destroy_process_group()


def test_counter_all_reduce_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, expect = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(total, expect)
    m = sharding.metrics_from_counters(total)
    assert abs(m["acc"] - (expect[1].sum() / expect[0].sum())) < 1e-12
    assert 0 < m["miou"] < 1 and 0 < m["micro_iou"] < 1
