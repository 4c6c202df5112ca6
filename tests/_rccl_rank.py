"""Helper of tests/test_gpu_rccl.py: ONE rank of an RCCL ("nccl" backend on ROCm) process group on the MI355X of the box.
A fresh process (torch imported here, nothing touched the GPU before): init_process_group("nccl", world_size=1) bound to
cuda:0, the int64 [3][13] segmentation counters of the product path (psg_seg_stats on synthetic log-probs) summed through
sharding.reduce_counters - the one collective of the whole framework -, a MAX all-reduce like bench.py's timer,
a barrier, destroy_process_group.  Writes what it saw as JSON."""
import json
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import _lib, runtime, sharding      # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if "MASTER_PORT" not in os.environ:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    rng = np.random.default_rng(3 + rank)
    rows = 8192
    logp = torch.from_numpy(rng.standard_normal((rows, 13)).astype(np.float32)).cuda()
    gt = torch.from_numpy(rng.integers(0, 13, rows).astype(np.int32)).cuda()
    counters = torch.zeros(3, 13, dtype=torch.int64, device="cuda")
    _lib.call("psg_seg_stats", runtime.ptr(logp), runtime.ptr(gt), rows, 13, runtime.ptr(counters), None, runtime.stream())
    local = counters.cpu().numpy().copy()
    total = sharding.reduce_counters(counters)
    t = torch.tensor([1.5 + rank], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ones = torch.ones(1, dtype=torch.float64, device="cuda")
    dist.all_reduce(ones)
    dist.barrier()
    torch.cuda.synchronize()
    res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "local": local.tolist(), "total": total.cpu().numpy().tolist(),
           "max": float(t.item()), "ranks_seen": int(ones.item()),
           "host_counters": sharding.seg_counters_host(logp.argmax(1).cpu().numpy(), gt.cpu().numpy()).tolist()}
    dist.destroy_process_group()
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(res, fh)


if __name__ == "__main__":
    main()
