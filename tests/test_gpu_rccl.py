"""RCCL on the hardware that exists: a one-rank "nccl" process group on the box's MI355X, in a fresh child process
(started as a child, not a re-exec; the child imports torch itself).  The framework's only collective - the all-reduce of
the int64 [3][13] segmentation counters (sharding.reduce_counters) - runs through RCCL for real, and so does
`python bench.py --gpus 1`, which since round 5 creates its process group at every N (n_ranks_seen is a collective)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_rccl_one_rank_counter_all_reduce(tmp_path):
    out = tmp_path / "rccl.json"
    r = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_rank.py"), str(out)], env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(out.read_text())
    assert res["backend"] == "nccl" and res["world"] == 1 and res["ranks_seen"] == 1 and res["max"] == 1.5
    local, total, host = (np.array(res[k], np.int64) for k in ("local", "total", "host_counters"))
    assert np.array_equal(local, host)                  # psg_seg_stats == the host statement of the counters
    assert np.array_equal(total, local)                 # one rank: the sum over ranks is the rank's own counters
    assert total[0].sum() == 8192


def test_bench_gpus1_runs_its_collectives_through_rccl():
    """bench.py at N = 1 takes the same code path as N > 1: the line's n_ranks_seen is an RCCL all-reduce."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-reference",
                        "--no-secondary", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["n_ranks_seen"] == 1
    assert line["config"]["process_group"] == {"backend": "nccl", "world": 1, "collective": "rccl"}, line["config"]["process_group"]
    assert line["value"] > 0
