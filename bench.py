#!/usr/bin/env python
"""Benchmark of the attack hot path: attacked rooms/s for NB non-targeted PGD on PointNet++ sem-seg (headline), with
the other BASELINE.json configurations as workloads of their own.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pointnet2|resgcn|tarnu|pointnet2_msg|randla]

One "step" = one attack call over one batch of synthetic S3DIS-shaped input (pointnet2: NB_attack eps=0.05, alpha=2/255,
40 PGD iterations on 8 rooms of 4096 points x 9 channels per GPU = BASELINE configs[1]), through the C ABI.  Inputs are
resident in HBM before the timed region.  Multi-GPU: one process per GPU; `python bench.py --gpus N` starts its own ranks
(torch.distributed.run children of a process that has not touched a GPU), or is started by torch.distributed.run; rooms
are sharded by rank with NO data-path collective, one RCCL all-reduce of int64 counters after the timed region.

Prints ONE JSON line (rank 0).  Every workload's line carries `roofline` (its dominant kernel, timed live with HIP events
on the launch stream in one extra profiled attack after the timed region; `traffic` from the committed PMC passes of the
same command) and, at N = 1, `cpu_baseline` (the CPU oracle on the host cores on a bounded sample; reported baseline
only).  The default (pointnet2, N = 1) line also carries the other workloads, shortened, under `secondary`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP runtime setting, before anything initialises the GPU: hardware queues per process (ROCm's default is 4).  Only the
# one-call-per-room NU protocol (--nu-mode per-room-calls: 12 host threads, one small attack each) has more than 4 streams
# with work; its throughput goes from 84 to ~120 rooms/s, the headline (3 streams) is unchanged (576.8 vs 576.9 rooms/s).
# Reported in the line as config.hip_runtime.
# (ROCm's default is 4; a library user who wants the one-call-per-room figure must export it too: INTEGRATION.md)
HIP_RUNTIME = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "16"), "rocm_default": "4",
               "set_by": "caller" if "GPU_MAX_HW_QUEUES" in os.environ else "bench.py"}
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

EPS, ALPHA, ITERS = 0.05, 2 / 255, 40
BATCH, NPOINT = 8, 4096
PEAK_FP32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0

# algorithmic MACs per row of each fused kernel (sum of cin*cout over its 1x1-conv layers) and rows per room
SA_DIMS = ((12, 32, 32, 64), (67, 64, 64, 128), (131, 128, 128, 256), (259, 256, 256, 512))
SA_ROWS = (1024 * 32, 256 * 32, 64 * 32, 16 * 32)
FP_DIMS = ((128, 128, 128, 128, 128, 13), (320, 256, 128), (384, 256, 256), (768, 256, 256))  # fp1+head, fp2, fp3, fp4
FP_ROWS = (4096, 1024, 256, 64)


def macs(dims):
    return sum(a * b for a, b in zip(dims[:-1], dims[1:]))


def kernel_flops(batch):
    """Algorithmic FLOPs per launch of the SSG modules (2 x MACs; the input-gradient pass has the same MAC count)."""
    out = {}
    for l in range(4):
        f = 2.0 * batch * SA_ROWS[l] * macs(SA_DIMS[l])
        out["sa%d_fwd" % (l + 1)] = f
        out["sa%d_bwd" % (l + 1)] = f
    names = ("fp1_head", "fp2", "fp3", "fp4")
    for l in range(4):
        f = 2.0 * batch * FP_ROWS[l] * macs(FP_DIMS[l])
        out[names[l] + "_fwd"] = f
        out[names[l] + "_bwd"] = f
    return out


def kernel_flops_executed(batch):
    """FLOPs the SSG launches EXECUTE since round 5 (the shipped defaults: PSG_PN2_SPLIT / PSG_PN2_FPSPLIT unset): the first layer
    of fp1-fp3 is split across the 3-NN interpolation (see the end of this function), and the first layer of SA levels 2-4 is split (psg_pn2_kernels.cuh, sa_fwd_kernel
    SPLIT) into a per-POINT feature product (pw_fwd / pw_bwd: N_l points, D -> C1) and a per-ROW xyz chunk (3 -> C1); the
    backward applies the first layer's transpose per point as well.  Levels: rows S_l x 32, points N_l = S_{l-1}.
    kernel_flops() stays the ALGORITHMIC count of the reference's layers (what `roofline` and the end-to-end figures use)."""
    out = kernel_flops(batch)
    pw = 0.0
    for l in (1, 2, 3):
        d_in, c1, c2, c3 = SA_DIMS[l]
        n_pts = SA_ROWS[l - 1] // 32
        out["sa%d_fwd" % (l + 1)] = 2.0 * batch * SA_ROWS[l] * (3 * c1 + c1 * c2 + c2 * c3)
        out["sa%d_bwd" % (l + 1)] = 2.0 * batch * SA_ROWS[l] * (c1 * c2 + c2 * c3)
        pw += 2.0 * batch * n_pts * (d_in - 3) * c1
    # level 0 of the attack loop's backward: three colour columns of the first layer (on the vector pipe) instead of twelve
    d_in, c1, c2, c3 = SA_DIMS[0]
    out["sa1_bwd"] = 2.0 * batch * SA_ROWS[0] * (3 * c1 + c1 * c2 + c2 * c3)
    out["pw_fwd"] = pw       # (three launches per forward, one per split level)
    out["pw_bwd"] = pw
    # FP split (fp_layer1_split): the interpolated-part columns of fp1-fp3's first layer run per COARSE point, as one more
    # layer of the coarser module (forward: behind its last layer; backward: in front of its first)
    names = ("fp1_head", "fp2", "fp3", "fp4")
    skip = (0, 64, 128, 256)                                  # C1: the skip-link channels of fp1..fp4's concatenated input
    for l in range(4):
        dims = FP_DIMS[l]
        m = macs(dims)
        if l < 3:
            m -= (dims[0] - skip[l]) * dims[1]                # this module's interpolated part leaves its kernels
        if l > 0:
            fine = FP_DIMS[l - 1]
            m += (fine[0] - skip[l - 1]) * fine[1]            # ... and is the extra layer of the next coarser module
        out[names[l] + "_fwd"] = out[names[l] + "_bwd"] = 2.0 * batch * FP_ROWS[l] * m
    return out


def kernel_flops_msg(batch):
    """The same for pointnet2_sem_seg_msg (the two scales of an SA level share one tag)."""
    from pointsecguard_amd.synthetic import MSG_FP, MSG_SA
    out = {}
    for l, ((cin, mlps), s_l) in enumerate(zip(MSG_SA, (1024, 256, 64, 16))):
        f = 2.0 * batch * sum(s_l * k * macs((cin + 3,) + tuple(mlp)) for mlp, k in zip(mlps, (16, 32)))
        out["sa%d_fwd" % (l + 1)] = f
        out["sa%d_bwd" % (l + 1)] = f
    for (name, cin, mlp), n_l in zip(MSG_FP, (64, 256, 1024, 4096)):
        dims = (cin,) + tuple(mlp) + ((128, 13) if name == "fp1" else ())
        tag = "fp1_head" if name == "fp1" else name
        f = 2.0 * batch * n_l * macs(dims)
        out[tag + "_fwd"] = f
        out[tag + "_bwd"] = f
    return out


def kernel_flops_msg_executed(batch):
    """FLOPs the MSG launches EXECUTE with the shipped defaults (round-5 review: the MSG line printed the algorithmic count
    only, although the same splits are active there): SA levels 2-4 run their first layer as a per-POINT feature product
    (pw_fwd / pw_bwd) + a per-ROW xyz chunk, per scale; fp1-fp3's interpolated first-layer columns run per COARSE point inside
    the coarser module; level 0 of the attack loop's backward applies three colour columns of the first layer (vector
    pipe) instead of all twelve.  Mirrors kernel_flops_executed."""
    from pointsecguard_amd.synthetic import MSG_FP, MSG_SA
    out = kernel_flops_msg(batch)
    S, KS = (1024, 256, 64, 16), (16, 32)
    pw = 0.0
    for l in (1, 2, 3):
        cin, mlps = MSG_SA[l]
        f = b = 0.0
        for (c1, c2, c3), k in zip(mlps, KS):
            f += S[l] * k * (3 * c1 + c1 * c2 + c2 * c3)
            b += S[l] * k * (c1 * c2 + c2 * c3)
            pw += S[l - 1] * cin * c1
        out["sa%d_fwd" % (l + 1)] = 2.0 * batch * f
        out["sa%d_bwd" % (l + 1)] = 2.0 * batch * b
    cin, mlps = MSG_SA[0]
    out["sa1_bwd"] = 2.0 * batch * sum(S[0] * k * (3 * c1 + c1 * c2 + c2 * c3) for (c1, c2, c3), k in zip(mlps, KS))
    out["pw_fwd"] = out["pw_bwd"] = 2.0 * batch * pw
    # FP split: modules in MSG_FP order fp4, fp3, fp2, fp1; skip-link channels of their concatenated inputs
    mods = {name: ((cin,) + tuple(mlp) + ((128, 13) if name == "fp1" else ()), n_l)
            for (name, cin, mlp), n_l in zip(MSG_FP, (64, 256, 1024, 4096))}
    skip = {"fp1": 0, "fp2": MSG_SA[1][0], "fp3": MSG_SA[2][0], "fp4": MSG_SA[3][0]}
    order = ("fp1", "fp2", "fp3", "fp4")
    for i, name in enumerate(order):
        dims, n_l = mods[name]
        m = macs(dims)
        if name != "fp4":
            m -= (dims[0] - skip[name]) * dims[1]              # this module's interpolated part leaves its kernels
        if i > 0:
            fine, _ = mods[order[i - 1]]
            m += (fine[0] - skip[order[i - 1]]) * fine[1]      # ... and is the extra layer of the next coarser module
        tag = "fp1_head" if name == "fp1" else name
        out[tag + "_fwd"] = out[tag + "_bwd"] = 2.0 * batch * n_l * m
    return out


def pn2_roofline(prof, flops, sa_launches_per_call=1):
    """roofline object of the MLP module with the largest total time in a HIP-event profile {tag: (ms, launches)}.
    (MSG network: an SA level is two launches, one per scale, under one tag; flops[tag] covers both.)"""
    mlp = {k: v for k, v in prof.items() if k in flops}
    dom = max(mlp, key=lambda k: mlp[k][0])
    n = mlp[dom][1] / (sa_launches_per_call if dom.startswith("sa") else 1)
    avg_ms = mlp[dom][0] / n
    achieved = flops[dom] / (avg_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": None, "avg_launch_us": avg_ms * 1e3, "launches": mlp[dom][1],
            "flop_per_launch": flops[dom]}


class Ranks:
    """One process per GPU (torch.distributed.run): RANK / LOCAL_RANK / WORLD_SIZE from the environment, RCCL
    (backend "nccl") for the barrier and the max-over-ranks time; PSG_BENCH_BACKEND=gloo lets several ranks share one
    GPU to rehearse the multi-rank path on a single-GPU box (RCCL refuses two ranks on one device).  Work is sharded
    by rank with no data-path collective: every rank attacks its own rooms / clouds (weak scaling).
    `cuda=False` (the launch rehearsal only, see main_rehearse) keeps every torch.cuda call out."""

    def __init__(self, args, cuda=True):
        import torch
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.cuda = cuda
        if args.gpus != self.world:
            raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: start it plainly (python bench.py --gpus N "
                             "launches its own ranks) or with torch.distributed.run --nproc-per-node N" % (args.gpus, self.world))
        backend = os.environ.get("PSG_BENCH_BACKEND", "nccl") if cuda else "gloo"
        self.device = "cuda" if cuda else "cpu"
        if cuda:
            self.dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(self.dev_index)
        self.dist = None
        self.pg_info = {"backend": None, "world": self.world, "collective": "none (--no-process-group)"}
        # A process group at EVERY N, N = 1 included (round 5): the barrier, the max-over-ranks time and n_ranks_seen of the
        # one-GPU line are then real RCCL collectives, the same code path the 2/4/8-GPU runs take.
        if self.world > 1 or (cuda and not getattr(args, "no_process_group", False)):
            import torch.distributed as dist
            if cuda and os.environ.get("PSG_BENCH_PG_FIRST") != "1":
                # torch's stream pool (32 streams per priority, all created at the first request) BEFORE the communicator's own
                # streams: the runtime deals streams to its hardware queues in creation order, and the launch streams of the
                # workloads keep the assignment they had without a process group (a stream created earlier shifts
                # everybody else's queue; measured on the MSG / tarnu lines, round 5)
                torch.cuda.Stream(device=self.dev_index)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            init_kw = {}
            if "MASTER_PORT" not in os.environ and self.world == 1:
                # plain `python bench.py` at N = 1: no launcher set a rendezvous, and none is needed across processes - a
                # file store in a private directory (round-5 advisor: picking a free TCP port by bind-then-close races when
                # several benches start together, as the tools/ sweeps do)
                import tempfile
                self._pg_dir = tempfile.mkdtemp(prefix="psg_bench_pg_")
                init_kw["init_method"] = "file://" + os.path.join(self._pg_dir, "store")
            try:
                if backend == "nccl":
                    dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.dev_index), **init_kw)
                else:
                    dist.init_process_group(backend, rank=self.rank, world_size=self.world, **init_kw)
                self.dist = dist
                self.pg_info = {"backend": backend, "world": dist.get_world_size(), "collective": "rccl" if backend == "nccl" else backend}
            except Exception as exc:
                if self.world > 1:
                    raise
                # one rank: the group only makes the barrier / max / n_ranks_seen real collectives; a box whose communicator
                # does not come up still gets its line, which says so
                self.pg_info = {"backend": backend, "world": 1, "collective": "none (init_process_group failed: %s: %s)" % (type(exc).__name__, str(exc)[:200])}

    def fence(self):
        import torch
        if self.cuda:
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if self.cuda:
            torch.cuda.synchronize()

    def timed(self, fn):
        """barrier + synchronize, fn(), barrier + synchronize; returns the MAX elapsed seconds over ranks."""
        import torch
        self.fence()
        t0 = time.perf_counter()
        fn()
        self.fence()
        elapsed = time.perf_counter() - t0
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def sum(self, x):
        import torch
        if self.dist is None:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t)
        return float(t.item())

    def done(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start the N ranks ourselves, one process per GPU, as
    children of THIS process (which has not imported torch, let alone touched a GPU: nothing is re-exec'ed), relay
    their output (rank 0 prints the JSON line) and exit with the launcher's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.Popen(cmd, env=env)
    raise SystemExit(proc.wait())


def main_rehearse(args):
    """PSG_BENCH_REHEARSE=1: the launch / rendezvous / barrier / max-over-ranks / rank-0-prints plumbing of every
    workload with NO device work (CPU test of `python bench.py --gpus 2`, gloo).  The line says so and carries no value."""
    R = Ranks(args, cuda=False)
    elapsed = R.timed(lambda: time.sleep(0.002 * args.steps * (1 + R.rank)))
    units = R.sum(args.steps)
    result = {"metric": "rehearsal", "rehearsal": True, "value": None, "n_gpus": R.world, "n_ranks_seen": int(R.sum(1)),
              "scaling": getattr(args, "scaling", "weak"), "steps": args.steps,
              "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "units_all_ranks": units,
              "workload": args.workload}
    if R.rank == 0:
        print(json.dumps(result), flush=True)
    R.done()
    return result


ENV_SWITCHES = []     # filled by check_env_switches(): what the line reports as config.env_switches


def check_env_switches(allow_paths, diag_ok=False):
    """The benchmark must not be able to skip its work by environment.  The library reports the switches it knows that are
    set (`psg_env_switches`, include/psg.h).  A library built with the work-skipping timing switches compiled in
    (-DPSG_DIAG_BUILD, tools/diag_*.sh) or a result-changing switch stops the run; a path-selecting switch ('p': another
    tested kernel path, same results) stops it too unless --allow-env-switches says the A/B run is intended; everything
    set is printed in the line as config.env_switches ([] = the shipped defaults)."""
    from pointsecguard_amd import _lib
    lib = _lib.load()
    sw = _lib.env_switches()
    for name in sorted(os.environ):
        if name.startswith("PSG_BENCH_"):
            sw.append((name, os.environ[name], "b"))
    if os.environ.get("PSG_LIBRARY_OVERRIDE"):
        sw.append(("PSG_LIBRARY_OVERRIDE", os.environ["PSG_LIBRARY_OVERRIDE"], "p"))     # a variant build of libpsg.so
    ENV_SWITCHES[:] = ["%s=%s (%s)" % s for s in sw]
    if lib.psg_diag_build() and diag_ok:
        ENV_SWITCHES.append("DIAGNOSTIC BUILD (-DPSG_DIAG_BUILD): value withheld")
        return
    if lib.psg_diag_build():
        raise SystemExit("bench.py: libpsg.so was built with -DPSG_DIAG_BUILD (kernels can skip work): not a benchmark library; "
                         "rebuild with `make -C pointsecguard_amd/csrc clean all`")
    bad = [s for s in sw if s[2] == "r" or (s[2] == "p" and not allow_paths)]
    if bad:
        raise SystemExit("bench.py: refusing to run with non-default library switches %s (use --allow-env-switches for an "
                         "A/B run of a tested alternative path; result-changing switches are never accepted)" % bad)


def base_line(metric, unit, value, R, args, elapsed, workload, extra_config=None):
    cfg = {"workload": workload, "sharding": "independent rooms / clouds sharded by rank, no data-path collective",
           "env_switches": list(ENV_SWITCHES), "hip_runtime": dict(HIP_RUNTIME), "process_group": dict(R.pg_info),
           # hipGraph bookkeeping of the replayed loops since the previous line of this process (psg_capture_stats): a failed
           # capture costs speed, not correctness, so it is counted here instead of raised
           "hipgraph": capture_delta()}
    if extra_config:
        cfg.update(extra_config)
    # n_ranks_seen: a sum of ones over the process group (RCCL on GPUs): evidence in the line itself that N ranks took part
    return {"metric": metric, "value": value, "unit": unit, "n_gpus": R.world, "n_ranks_seen": int(R.sum(1)), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg}


_CAPTURE_MARK = {}


def capture_delta():
    """{captures_tried, captures_failed, replays, eager} of the library's replayed loops since the last call."""
    try:
        from pointsecguard_amd import _lib
        now = _lib.capture_stats()
    except Exception as exc:            # (the launch rehearsal has no library)
        return {"error": str(exc)}
    d = {k: v - _CAPTURE_MARK.get(k, 0) for k, v in now.items()}
    _CAPTURE_MARK.update(now)
    return d


def cpu_threads():
    """Threads the CPU-oracle sample actually runs on: torch's intra-op pool, which main() caps at the rank's share of the
    host cores (the oracle's C parts are single-threaded per call, its tensor parts use this pool)."""
    import torch
    return torch.get_num_threads()


def want_cpu(args, R):
    return R.rank == 0 and R.world == 1 and not args.no_cpu_baseline      # reported baseline: rank 0 at N = 1 only


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="default workload: skip the shortened other workloads")
    ap.add_argument("--cpu-iters", type=int, default=6, help="pointnet2: PGD iterations of the CPU-oracle sample")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="other workloads: rough budget of the CPU-oracle sample")
    ap.add_argument("--no-reference", action="store_true",
                    help="skip the uncoalesced reference runs (profiling passes: keeps every launch of a kernel "
                         "at the same device batch, so per-launch averages mean something)")
    ap.add_argument("--workload", default="pointnet2", choices=["pointnet2", "resgcn", "tarnu", "pointnet2_msg", "randla"],
                    help="pointnet2 = BASELINE configs[1] (headline metric); resgcn = configs[3] (ResGCN-28); tarnu = "
                         "configs[2] (targeted NU attack, batch 32); randla = configs[4]; pointnet2_msg = the headline "
                         "attack on the multi-scale-grouping network (SURVEY 8f rank 2)")
    ap.add_argument("--coalesce", type=int, default=8,
                    help="consecutive steps (batches of 8 rooms) fused into one device batch per launch; rooms are "
                         "independent, so results are identical and small kernels get more workgroups")
    ap.add_argument("--nu-steps", type=int, default=0,
                    help="tarnu workload: optimiser step cap per attack (0 = 40 per room, SURVEY 8(d)(3); 100 for --nu-mode batch32)")
    ap.add_argument("--nu-mode", default="per-room", choices=["per-room", "per-room-calls", "batch32"],
                    help="tarnu workload: the attack applied per room (the reference's batch-of-one semantics; default: the rooms "
                         "of a step advanced in lockstep, per-room-calls: one call per room) or one call on the whole 32-room "
                         "batch (its exit test then fires after one step)")
    ap.add_argument("--nu-coalesce", type=int, default=2,
                    help="tarnu workload, per-room mode: consecutive 32-room steps advanced by one lockstep call")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="tarnu workload: weak = 32 rooms per step on EVERY GPU; strong = the 32 rooms of a step split over the "
                         "GPUs (BASELINE configs[2]: 'batch=32 rooms, sharded 8x' = 4 rooms per GPU)")
    ap.add_argument("--randla-iters", type=int, default=100,
                    help="randla workload: BIM iterations per attacked cloud (BASELINE configs[4]: 100)")
    ap.add_argument("--randla-metric", default="l_2", choices=["l_2", "l_inf"],
                    help="randla workload: distance metric of the BIM update.  l_2 (default) is what the reference's tester runs "
                         "(RandLA-Net/tester_S3DIS.py:37,142-145: NBattack(..., distance_metric='l_2'), magnitude 17, alpha 1.7); "
                         "l_inf (eps 0.05, alpha 0.01) is reported beside it as `l_inf`")
    ap.add_argument("--nu-plan-ahead", type=int, default=0,
                    help="tarnu workload: forwards per geometry plan (multiple of 10; 0 = the package's default, nu.plan_ahead)")
    ap.add_argument("--nu-concurrency", type=int, default=12,
                    help="tarnu workload: attacks in flight, one host thread + HIP stream + model instance each (a one-room "
                         "attack is ~30 short launches per optimiser step: several of them side by side fill the GPU)")
    ap.add_argument("--gcn-concurrency", type=int, default=4, help="resgcn workload: attacks in flight (streams; 3 -> 4: 14.75 -> 14.93, tools/r05_o.sh)")
    ap.add_argument("--gcn-block", default="res", choices=["res", "plain", "dense"],
                    help="resgcn workload: backbone block (architecture.py:26-39 of the reference); default = BASELINE's")
    ap.add_argument("--gcn-conv", default="edge", choices=["edge", "mr"], help="resgcn workload: graph convolution")
    ap.add_argument("--gcn-blocks", type=int, default=28, help="resgcn workload: number of blocks (BASELINE: 28)")
    ap.add_argument("--gcn-batch", type=int, default=1, help="resgcn workload: rooms per step (the reference's loader: 1)")
    ap.add_argument("--gcn-coalesce", type=int, default=4,
                    help="resgcn workload: steps fused into one device batch per launch (rooms are independent; the CE "
                         "mean's scale changes by an exact power of two, which sign() ignores)")
    ap.add_argument("--rla-coalesce", type=int, default=12,
                    help="randla workload: steps (clouds) fused into one device batch per launch; the clouds stay "
                         "independent (every index stays inside its cloud), one launch of each kernel serves all of them")
    ap.add_argument("--rla-concurrency", type=int, default=4,
                    help="randla workload: launches in flight (12 clouds x 4 and 16 x 4 measured 20.1 clouds/s, 16 x 3 19.85, "
                         "24 x 3 20.0, 24 x 2 18.8: tools/r05_m.sh)")
    ap.add_argument("--concurrency", type=int, default=3,
                    help="device batches in flight per GPU (one HIP stream + workspace each)")
    ap.add_argument("--no-process-group", action="store_true",
                    help="N = 1 only, profiler passes: do not create the one-rank RCCL process group (its kernels and threads stay "
                         "out of counter collection); the line then says config.process_group.collective = none")
    ap.add_argument("--allow-env-switches", action="store_true",
                    help="accept PSG_* switches that select another TESTED code path (A/B runs); they are listed in "
                         "config.env_switches either way, result-changing ones are always refused")
    ap.add_argument("--diag-build-ok", action="store_true",
                    help="tools/diag_*.sh only: run on a -DPSG_DIAG_BUILD library for its per-kernel times; the line then "
                         "carries value = null and invalid = 'diagnostic build'")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    if os.environ.get("PSG_BENCH_REHEARSE") == "1":
        return main_rehearse(args)
    check_env_switches(args.allow_env_switches, args.diag_build_ok)
    R = Ranks(args)
    import torch
    # torch's CPU thread pool: the box gives a rank a share of the host cores (16 per GPU), torch sizes its pool by the
    # machine (128+): every small CPU tensor op of the host-side API code then costs milliseconds of oversubscribed wake-ups
    torch.set_num_threads(max(1, min(torch.get_num_threads(), 16, (os.cpu_count() or 16) // max(1, R.world))))
    runners = {"pointnet2": run_pointnet2, "resgcn": run_resgcn, "tarnu": run_tarnu, "pointnet2_msg": run_msg,
               "randla": run_randla}
    result = runners[args.workload](args, R)
    if args.workload == "pointnet2" and R.world == 1 and not args.no_secondary:
        # the other BASELINE configurations, shortened so that the default run stays within a few minutes: each a complete
        # line of its own (value, roofline, cpu_baseline) under "secondary"; `python bench.py --workload NAME` runs one alone
        import copy
        sec = {}
        for name, steps, warm in (("tarnu", 16, 8), ("resgcn", 24, 8), ("pointnet2_msg", 24, 8), ("randla", 48, 12)):
            a2 = copy.copy(args)
            a2.workload, a2.steps, a2.warmup, a2.cpu_seconds = name, steps, warm, min(args.cpu_seconds, 6.0)
            if name == "resgcn":
                # behind the headline's and the NU line's streams a fourth ResGCN launch in flight costs more than it gives
                # (14.2 against 14.7 - 14.8 rooms/s; alone: 14.9 with four, 14.75 with three): three here, four in the line of its own
                a2.gcn_concurrency = min(args.gcn_concurrency, 3)
            t0 = time.time()
            try:
                sec[name] = runners[name](a2, R)
                sec[name]["bench_wall_s"] = round(time.time() - t0, 1)
            except Exception as exc:                               # a secondary line must not take the headline down
                import traceback
                sec[name] = {"error": "%s: %s" % (type(exc).__name__, exc), "traceback": traceback.format_exc()[-1500:]}
        result["secondary"] = sec
    if args.diag_build_ok and any(e.startswith("DIAGNOSTIC BUILD") for e in ENV_SWITCHES):
        result["value"], result["invalid"] = None, "diagnostic build: kernels may have skipped work"
    if R.rank == 0:
        print(json.dumps(result), flush=True)
    R.done()
    return result


# ======================================================================================== pointnet2 (headline)
def cpu_baseline_pn2(sd, rooms, labels, starts, iters_sample, msg=False):
    """The CPU oracle (oracle/, a port of the reference algorithm) on the host cores; bounded sample."""
    from oracle import attacks as oatk
    from oracle import pn2
    if msg:
        from oracle import pn2_msg
        orc = pn2_msg.PN2MsgOracle(sd)
    else:
        orc = pn2.PN2Oracle(sd)
    images = np.ascontiguousarray(rooms.transpose(0, 2, 1))
    t0 = time.time()
    oatk.nb_attack(orc, images, labels, EPS, ALPHA, iters_sample, starts[:iters_sample])
    dt = time.time() - t0
    per_room_attack = dt / iters_sample * ITERS / rooms.shape[0]
    return {"value": 1.0 / per_room_attack, "unit": "attacked rooms/s", "cores": cpu_threads(), "kind": "port",
            "sample": "%d rooms x %d of %d PGD iterations in %.1f s, extrapolated linearly" % (
                rooms.shape[0], iters_sample, ITERS, dt)}


def run_pointnet2(args, R):
    import torch
    from pointsecguard_amd import runtime
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    rank, world, dist = R.rank, R.world, R.dist

    sd = dict(np.load(os.path.join(ROOT, "tests", "golden", "pn2_weights.npz")))
    model = runtime.PN2Model(runtime.fold_state_dict(sd))
    # K steps of BATCH rooms are coalesced up to G at a time into device batches (launches); when G does not divide K the
    # K steps are dealt evenly to ceil(K / G) launches (20 steps -> 7, 7, 6 rather than 8, 8, 4: launches that run side
    # by side then finish together).  Warm-up runs whole device batches (>= W steps, untimed).
    G = max(1, min(args.coalesce, args.steps))                                    # steps per device batch, at most
    DB = BATCH * G                                                                # rooms per full launch
    n_l = -(-args.steps // G)
    sizes = [args.steps // n_l + (1 if i < args.steps % n_l else 0) for i in range(n_l)]
    n_warm = -(-args.warmup // G) if args.warmup > 0 else 0
    sizes = [G] * n_warm + sizes                                                  # steps in device batch i
    n_all, n_groups = len(sizes), len(sizes) - n_warm
    conc = max(1, min(args.concurrency, n_groups))
    streams = [torch.cuda.Stream() for _ in range(conc)]
    ws_cache = {}

    def workspace(slot, rooms_in_batch):
        key = (slot, rooms_in_batch)
        if key not in ws_cache:
            ws_cache[key] = runtime.PN2Workspace(rooms_in_batch, NPOINT, ITERS)
        return ws_cache[key]

    ws = workspace(0, DB)

    # each rank attacks its own shard of rooms (weak scaling: BATCH rooms per GPU per step)
    rooms = [make_rooms(BATCH * sizes[s], 1000 + rank * 100003 + s) for s in range(n_all)]
    labels = [rule_labels(r) for r in rooms]
    # FPS start draws: the columns of this rank's rooms in the table ONE process would draw for the global batch of the
    # step (every rank seeds the same generator), so the union over ranks is the single-process stream
    from pointsecguard_amd import sharding
    starts = [sharding.fps_start_table_sharded(1234 + s, ITERS, BATCH * sizes[s] * world, rank * BATCH * sizes[s],
                                               (rank + 1) * BATCH * sizes[s], NPOINT) for s in range(n_all)]
    d_images = [torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 1))).cuda() for r in rooms]
    d_labels = [torch.from_numpy(l.astype(np.int32)).cuda() for l in labels]
    d_starts = [torch.from_numpy(s).cuda() for s in starts]
    d_adv = [torch.empty_like(x) for x in d_images]
    for i in range(n_all):
        workspace(i % conc, BATCH * sizes[i])                                     # allocate outside the timed region

    def step(i):
        # device batch i runs on stream i % conc with its own workspace: batches are independent, so kernels of
        # one overlap with kernels of the others (no cross-step dependency, no collective)
        with torch.cuda.stream(streams[i % conc]):
            workspace(i % conc, BATCH * sizes[i]).nb_attack(model, d_images[i], d_labels[i], d_starts[i], EPS, ALPHA,
                                                            ITERS, out=d_adv[i])

    for i in range(n_warm):
        step(i)
    elapsed = R.timed(lambda: [step(i) for i in range(n_warm, n_all)])

    # ---- attack statistics over the timed steps: clean vs adversarial accuracy / mIoU (RCCL all-reduce of counters)
    clean = torch.zeros(3, 13, dtype=torch.int64, device="cuda")
    adv = torch.zeros(3, 13, dtype=torch.int64, device="cuda")
    evs = {}
    for i in range(n_warm, min(n_all, n_warm + 2)):
        ev = evs.setdefault(sizes[i], runtime.PN2Workspace(BATCH * sizes[i], NPOINT, 1))
        ev_starts = d_starts[i][:1].contiguous()
        for src, ctr in ((d_images[i], clean), (d_adv[i], adv)):
            x0 = src.transpose(1, 2).contiguous()
            ev.plan_build(x0, ev_starts, 1)
            runtime.seg_stats(ev.forward(model, 0, x0), d_labels[i], counters=ctr)
    if dist is not None:
        dist.all_reduce(clean)
        dist.all_reduce(adv)
    clean, adv = clean.cpu().numpy().astype(np.float64), adv.cpu().numpy().astype(np.float64)
    acc, adv_acc = clean[1].sum() / clean[0].sum(), adv[1].sum() / adv[0].sum()
    miou = float(np.mean((clean[1] / (clean[2] + 1e-6))[clean[0] != 0]))
    adv_miou = float(np.mean((adv[1] / (adv[2] + 1e-6))[adv[0] != 0]))

    result = base_line("attacked rooms/sec (4096 pts, 40 PGD iters)", "rooms/s", BATCH * world * args.steps / elapsed, R, args,
                       elapsed, "NB non-targeted PGD (eps=0.05, alpha=2/255, 40 iters) on PointNet++ SSG sem_seg, batch=8 "
                       "rooms x 4096 pts x 9 ch per GPU (BASELINE configs[1])",
                       {"rooms_per_step_per_gpu": BATCH, "steps_coalesced_per_launch_max": G,
                        "device_batch_rooms": BATCH * max(sizes[n_warm:]),
                        "launch_sizes_in_steps": sizes[n_warm:], "launches_in_flight_per_gpu": conc,
                        "weights": "tests/golden/pn2_weights.npz (fitted fixture)"})
    if rank == 0:
        # ---- for reference: the same attack launched one 8-room step at a time (no coalescing), 2 in flight
        ref8 = None
        if G > 1 and not args.no_reference:
            n8 = int(os.environ.get("PSG_BENCH_STRICT_INFLIGHT", "4"))
            st8 = [torch.cuda.Stream() for _ in range(n8)]
            ws8 = [runtime.PN2Workspace(BATCH, NPOINT, ITERS) for _ in range(n8)]
            g8 = sizes[n_warm]                                                    # steps of the first timed device batch
            x8 = [d_images[n_warm][i * BATCH:(i + 1) * BATCH].contiguous() for i in range(g8)]
            l8 = [d_labels[n_warm][i * BATCH:(i + 1) * BATCH].contiguous() for i in range(g8)]
            s8 = [d_starts[n_warm][:, :, i * BATCH:(i + 1) * BATCH].contiguous() for i in range(g8)]
            o8 = [torch.empty_like(x) for x in x8]

            def run8():
                for i in range(len(x8)):
                    with torch.cuda.stream(st8[i % n8]):
                        ws8[i % n8].nb_attack(model, x8[i], l8[i], s8[i], EPS, ALPHA, ITERS, out=o8[i])
                torch.cuda.synchronize()

            run8()
            t8 = time.perf_counter()
            run8()
            ref8 = {"value": BATCH * len(x8) / (time.perf_counter() - t8), "unit": "rooms/s",
                    "note": "one launch per 8-room step, no coalescing, %d steps in flight (this GPU only)" % n8}
            del ws8
        # ---- roofline of the dominant kernel: one extra attack with HIP-event timing of every launch
        rooms_r = BATCH * sizes[n_warm]                                           # the first timed device batch
        wsr = workspace(0, rooms_r)
        wsr.prof_enable(True)
        wsr.nb_attack(model, d_images[n_warm], d_labels[n_warm], d_starts[n_warm], EPS, ALPHA, ITERS, out=d_adv[n_warm])
        torch.cuda.synchronize()
        prof = wsr.prof_read()
        wsr.prof_enable(False)
        flops = kernel_flops(rooms_r)
        fx = kernel_flops_executed(rooms_r)
        roof = pn2_roofline(prof, flops)
        # `achieved` / `frac` count the ALGORITHMIC FLOPs of the reference's layers (the contract's definition); since round 5 the
        # launch EXECUTES fewer (fp1 + head: four of its five layers - the first one runs per coarse point inside fp2's kernel),
        # and the matrix-busy counters (profiles/*_pmc_mfma.txt) measure what is executed: both are in the line
        roof["flop_per_launch_executed"] = fx[roof["kernel"]]
        roof["frac_executed"] = fx[roof["kernel"]] / (roof["avg_launch_us"] * 1e-6) / 1e12 / PEAK_FP32_MATRIX_TFLOPS
        roof["rooms_per_launch"] = rooms_r
        roof["traffic"], roof["traffic_source"] = pmc_traffic(roof["kernel"], rooms_r)
        roof["algorithmic_bytes"] = FP1_FWD_BYTES_PER_ROOM * rooms_r if roof["kernel"] == "fp1_head_fwd" else None
        total_ms = sum(v[0] for v in prof.values())
        result.update({
            "roofline": roof,
            "kernel_ms_per_attack": {k: round(v[0], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
            "kernel_ms_total_per_attack": round(total_ms, 3),
            # per kernel: EXECUTED FLOPs / time / peak (since round 5 the SA levels 2-4 execute fewer MACs than the reference's
            # layers count: kernel_flops_executed); pw_* are three launches per pass under one tag
            "mfma_frac_by_kernel": {k: round(fx[k] / (v[0] / (v[1] / (3 if k.startswith("pw_") else 1)) * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 3)
                                    for k, v in prof.items() if k in fx},
            "executed_flop_share": round(sum(fx.values()) / sum(flops.values()), 4),
            "uncoalesced_reference": ref8,
            "parity": {"clean_acc": acc, "adv_acc": adv_acc, "asr": 1.0 - adv_acc, "clean_miou": miou,
                       "adv_miou": adv_miou, "rooms_evaluated": int(clean[0].sum() // NPOINT)},
        })
        if not args.no_reference:
            result["api_level"] = api_level_pn2(sd, d_images[n_warm], labels[n_warm])
            result["whole_scene"] = whole_scene_pn2(sd)
        if want_cpu(args, R):
            result["cpu_baseline"] = cpu_baseline_pn2(sd, rooms[n_warm][:BATCH], labels[n_warm][:BATCH],
                                                      starts[n_warm][:, :, :BATCH], args.cpu_iters)
    return result


def _reference_api_model(sd):
    import torch
    from pointsecguard_amd.models.pointnet2_sem_seg import get_model
    net = get_model(13)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.cuda().eval()


def api_level_pn2(sd, d_images, labels, n_calls=12):
    """The configs[1] attack EXACTLY as the reference's harness calls it (PointNet/NB_nontarget_test_semseg.py:163-171):
    `torchattacks.NB_attack(classifier, eps, alpha, iters)(images.cuda(), labels)` on one 8-room batch per call, labels as
    float64 numpy, from ONE host thread on the current stream - the 160 host `torch.randint` draws of the FPS starts, their
    upload, the label conversion and the packed-weight check included.  Calls are stream-ordered (nothing synchronises
    inside), so the host prepares call i + 1 while the GPU runs call i."""
    import torch
    from pointsecguard_amd.attacks import torchattacks
    net = _reference_api_model(sd)
    atk = torchattacks.NB_attack(net, EPS, ALPHA, ITERS)
    n_b = d_images.shape[0] // BATCH
    xs = [d_images[i * BATCH:(i + 1) * BATCH].contiguous() for i in range(n_b)]
    ys = [labels[i * BATCH:(i + 1) * BATCH].astype(np.float64) for i in range(n_b)]
    torch.manual_seed(0)
    out = atk(xs[0], ys[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_calls):
        out = atk(xs[i % n_b], ys[i % n_b])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del out
    # the same launches without the API layer: pre-drawn device-resident starts and int32 labels, the workspace called directly
    from pointsecguard_amd import runtime
    from pointsecguard_amd.models.pointnet2_sem_seg import draw_fps_starts
    ws = net._workspace(BATCH, NPOINT, ITERS)
    st_d = draw_fps_starts(BATCH, NPOINT, ITERS).cuda()
    ys_d = [torch.from_numpy(y.astype(np.int32)).cuda() for y in ys]
    o = torch.empty_like(xs[0])
    ws.nb_attack(net._packed(), xs[0], ys_d[0], st_d, EPS, ALPHA, ITERS, out=o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_calls):
        ws.nb_attack(net._packed(), xs[i % n_b], ys_d[i % n_b], st_d, EPS, ALPHA, ITERS, out=o)
    torch.cuda.synchronize()
    dt_c = time.perf_counter() - t0
    return {"value": BATCH * n_calls / dt, "unit": "rooms/s", "calls": n_calls,
            "c_abi_same_shape": BATCH * n_calls / dt_c, "api_overhead": 1.0 - dt_c / dt,
            "note": "torchattacks.NB_attack(get_model(13).cuda().eval(), 0.05, 2/255, 40)(x, y), 8 rooms per call, one host thread, "
                    "one stream, host RNG draws + uploads included (NB_nontarget_test_semseg.py:169-171); c_abi_same_shape = psg_pn2_nb_attack "
                    "called the same way (one 8-room attack at a time on one stream) with device-resident inputs"}


def whole_scene_pn2(sd, n_scenes=68, iters=10):
    """The reference's evaluation driver end to end (NB_nontarget_test_semseg.py:126-291 = harness.evaluate_whole_scene):
    block slicing of whole scenes on the host, clean forward, the harness's attack (NB_attack eps 0.1, alpha 0.05, 10
    iterations, :169), adversarial forward, vote pools, per-batch counters + L2 + TSV row (their read-backs included), per
    scene IoU - on a synthetic Area_5 of 68 scenes (the count of the real one, :117) of 4 m x 3 m each."""
    import tempfile
    import torch
    from pointsecguard_amd import harness
    from pointsecguard_amd.attacks import torchattacks
    rng = np.random.default_rng(5)
    scenes = {}
    for i in range(n_scenes):
        n = 60000
        xyz = rng.random((n, 3)) * np.array([4.0, 3.0, 2.8])
        scenes["Area_5_room_%d.npy" % i] = np.concatenate([xyz, np.floor(rng.random((n, 3)) * 256.0),
                                                            rng.integers(0, 13, n).astype(np.float64)[:, None]], axis=1)
    ds = harness.ScannetDatasetWholeScene(None, block_points=NPOINT, scenes=scenes)
    net = _reference_api_model(sd)
    np.random.seed(1)
    torch.manual_seed(1)
    n_blocks = [0]

    class Counting:                                   # counts the blocks the loop attacked (the dataset pads per column)
        def __init__(self, m):
            self.atk = torchattacks.NB_attack(m, eps=0.1, alpha=0.05, iters=iters)

        def __call__(self, x, y):
            n_blocks[0] += x.shape[0]
            return self.atk(x, y)

    with tempfile.TemporaryDirectory() as tmp:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = harness.evaluate_whole_scene(net, ds, Counting, batch_size=BATCH, num_votes=1, log_path=os.path.join(tmp, "log.txt"),
                                           log=lambda *_: None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"value": n_blocks[0] / dt, "unit": "rooms/s (4096-point blocks, %d PGD iterations each)" % iters, "scenes": n_scenes,
            "blocks": n_blocks[0], "seconds": dt, "adv_accuracy": res["adv_accuracy"], "accuracy": res["accuracy"],
            "note": "harness.evaluate_whole_scene: host block slicing, clean + adversarial forwards, NB_attack(eps=0.1, alpha=0.05, "
                    "iters=%d) as NB_nontarget_test_semseg.py:169, vote pools, per-batch read-backs and TSV rows" % iters}


# compulsory HBM bytes of the fp1 + head forward per room: coarse features [1024][128] fp32 + 3-NN indices and weights
# [4096][3] x 8 B in; log-probs [4096][13] fp32 + four layers of ReLU bits (16 bits per lane and 32x32 tile) out
FP1_FWD_BYTES_PER_ROOM = 1024 * 128 * 4 + 4096 * 3 * 8 + 4096 * 13 * 4 + 4 * (4096 // 32) * 4 * 64 * 2
# compulsory HBM bytes of one fused feature-space kNN launch per room: features in operand order [4096][64] + norms in,
# neighbour table [4096][16] int32 out (every workgroup re-reads the features from L2, not from HBM)
KNN_BYTES_PER_ROOM = 4096 * 64 * 4 + 4096 * 4 + 4096 * 16 * 4

# HIP symbols of the kernels whose instantiation is unique (PMC rows are keyed by symbol, not by module)
PMC_SYMBOL = {"fp1_head_fwd": "void psg::fp_fwd_kernel<32, 4, false>(psg::FpFwdArgs)",
              "fp1_head_bwd": "void psg::fp_bwd_kernel<32, 4, 1, false>(psg::FpBwdArgs)",
              "sa1_fwd": "void psg::sa_fwd_kernel<128, 4, 32, 1>(psg::SaFwdArgs)",
              "sa1_bwd": "void psg::sa_bwd_kernel<128, 4, 1, 32>(psg::SaBwdArgs)",
              "sa2_fwd": "void psg::sa_fwd_kernel<64, 4, 32, 1>(psg::SaFwdArgs)",
              "sa2_bwd": "void psg::sa_bwd_kernel<64, 4, 2, 32>(psg::SaBwdArgs)",
              # the fused kNN launches of the ResGCN path are two kernels since round 3 (the bf16-prefilter kernel for dilations
              # 1..20 in rounds 4-5 and for all of 1..27 since round 6, the exact kernel above): the tag's figure is the launch-weighted mean of both
              "knn_fused": ("(anonymous namespace)::knn_bf_kernel((anonymous namespace)::KnnBfArgs)",
                            "(anonymous namespace)::knn_fused_kernel((anonymous namespace)::KnnFusedArgs)")}
# MSG network: an SA level is TWO launches (one per scale) under one tag - the tag's traffic is their SUM per call (psg_pn2.hip:
# PSG_SA_BWD_CASE / the sa_fwd switch name the instantiations; sa3 / sa4 forward share their scale-0 instantiation, so only
# the backward tags and the FP modules are listed)
PMC_SYMBOL_MSG = {"sa1_bwd": ("void psg::sa_bwd_kernel<128, 4, 1, 16>(psg::SaBwdArgs)", "void psg::sa_bwd_kernel<128, 4, 1, 32>(psg::SaBwdArgs)"),
                  "sa2_bwd": ("void psg::sa_bwd_kernel<64, 4, 2, 16>(psg::SaBwdArgs)", "void psg::sa_bwd_kernel<64, 8, 1, 32>(psg::SaBwdArgs)"),
                  "sa3_bwd": ("void psg::sa_bwd_kernel<32, 8, 2, 16>(psg::SaBwdArgs)", "void psg::sa_bwd_kernel<32, 8, 2, 32>(psg::SaBwdArgs)"),
                  "sa4_bwd": ("void psg::sa_bwd_kernel<32, 8, 3, 16>(psg::SaBwdArgs)", "void psg::sa_bwd_kernel<32, 8, 3, 32>(psg::SaBwdArgs)"),
                  "sa1_fwd": ("void psg::sa_fwd_kernel<128, 4, 16, 1, false>(psg::SaFwdArgs)", "void psg::sa_fwd_kernel<128, 4, 32, 1, false>(psg::SaFwdArgs)"),
                  "sa2_fwd": ("void psg::sa_fwd_kernel<64, 4, 16, 1, true>(psg::SaFwdArgs)", "void psg::sa_fwd_kernel<64, 8, 32, 1, true>(psg::SaFwdArgs)"),
                  "fp1_head_fwd": "void psg::fp_fwd_kernel<32, 4, false>(psg::FpFwdArgs)",
                  "fp1_head_bwd": "void psg::fp_bwd_kernel<32, 4, 1, false>(psg::FpBwdArgs)"}


def pmc_traffic(tag, device_batch, pattern="*_pmc_traffic.json", symbols=None, combine="mean"):
    """HBM bytes per launch of kernel `tag` from the newest committed PMC summary (profiles/<round>_pmc_traffic*.json,
    written by tools/profile_round.sh: separate FETCH_SIZE / WRITE_SIZE passes of this bench at the default device
    batch, gfx950 read-doubling correction applied).  PMC cannot be sampled from inside the process, so the
    figure is the committed one (scaled by the room count when this run's device batch differs from the counted one)."""
    import glob
    symbols = PMC_SYMBOL if symbols is None else symbols
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files or tag not in symbols:
        return None, None
    with open(files[-1]) as fh:
        table = json.load(fh)
    sym = symbols[tag]
    if isinstance(sym, tuple):
        rows = [table[x] for x in sym if x in table]
        n = sum(r["launches_fetch_pass"] for r in rows)
        if combine == "sum":        # one call of the tag = one launch of EACH symbol (the two scales of an MSG SA level)
            row = {"hbm_bytes_per_launch": sum(r["hbm_bytes_per_launch"] for r in rows)} if len(rows) == len(sym) else None
        else:
            row = {"hbm_bytes_per_launch": sum(r["hbm_bytes_per_launch"] * r["launches_fetch_pass"] for r in rows) / n} if n else None
    else:
        row = table.get(sym)
    if not row:
        return None, None
    counted = table.get("_meta", {}).get("device_batch_rooms", 32)
    src = "profiles/" + os.path.basename(files[-1])
    if counted == device_batch:
        return row["hbm_bytes_per_launch"], src
    # the module kernels move a fixed volume per room (one workgroup tile per 32 points, weights from L2): a pass counted
    # at another device batch is scaled by the room count and says so
    return row["hbm_bytes_per_launch"] * device_batch / counted, src + " (counted at %d rooms per launch, scaled to %d)" % (counted, device_batch)


# ======================================================================================== resgcn (configs[3])
def run_resgcn(args, R):
    """BASELINE configs[3]: ResGCN-28 dense sem_seg forward/backward + non-targeted NB PGD, kNN k=16, 4096 points,
    one MI355X; the harness values eps=0.3, alpha=2/255, 50 iterations (ResGCN/sem_seg_dense/attacks.py:134).  A step is
    one attack call on --gcn-batch rooms (the reference's loader: 1); --gcn-coalesce consecutive steps share one device
    launch and --gcn-concurrency launches are in flight.  Weights: the fitted fixture of the BASELINE-size parity test."""
    import torch
    from pointsecguard_amd import runtime
    from pointsecguard_amd.synthetic import gcn28_state_dict, gcn_state_dict, make_rooms, rule_labels
    n_blocks, iters, batch = args.gcn_blocks, 50, args.gcn_batch
    default_cfg = (args.gcn_block, args.gcn_conv, n_blocks) == ("res", "edge", 28)
    sd = gcn28_state_dict() if default_cfg else gcn_state_dict(7, n_blocks, args.gcn_block, args.gcn_conv)
    F = 64 * n_blocks
    BLOCKS = {"res": runtime.GCN_BLOCK_RES, "plain": runtime.GCN_BLOCK_PLAIN, "dense": runtime.GCN_BLOCK_DENSE}
    CONVS = {"edge": runtime.GCN_CONV_EDGE, "mr": runtime.GCN_CONV_MR}
    cfg = dict(block=BLOCKS[args.gcn_block], conv=CONVS[args.gcn_conv])
    model = runtime.GCNModel(sd, n_blocks, **cfg)
    G = max(1, min(args.gcn_coalesce, args.steps))
    DB = batch * G
    n_launch, n_warm = -(-args.steps // G), (-(-args.warmup // G) if args.warmup > 0 else 0)
    conc = max(1, min(args.gcn_concurrency, n_launch))
    wss = [runtime.GCNWorkspace(DB, NPOINT, n_blocks, **cfg) for _ in range(conc)]
    streams = [torch.cuda.Stream() for _ in range(conc)]
    n_all = n_launch + n_warm
    rooms = [make_rooms(DB, 5000 + 1000 * R.rank + s) for s in range(n_all)]
    d_images = [torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 1))).cuda() for r in rooms]
    d_labels = [torch.from_numpy(rule_labels(r).astype(np.int32)).cuda() for r in rooms]
    d_adv = [torch.empty_like(x) for x in d_images]

    def launch(i, ws_list=wss, imgs=d_images, labs=d_labels, outs=d_adv):
        with torch.cuda.stream(streams[i % conc]):
            ws_list[i % conc].nb_attack(model, imgs[i], labs[i], 0.3, 2 / 255, iters, out=outs[i])

    for i in range(n_warm):
        launch(i)
    elapsed = R.timed(lambda: [launch(i) for i in range(n_warm, n_all)])
    rooms_done = DB * n_launch * R.world
    # algorithmic FLOPs per PGD iteration per room (kNN distance GEMMs + split EdgeConv + fusion/prediction + transposes)
    n = NPOINT
    gmac = (27 * n * n * 64 + n * n * 3 + 28 * n * 64 * 128 * 2 + n * F * 1024 + n * F * 512 * 2 + n * 512 * 256 * 2 +
            n * 256 * 13 * 2) / 1e9
    name = "ResGCN-%d" % n_blocks + ("" if (args.gcn_block, args.gcn_conv) == ("res", "edge") else
                                      " block=%s conv=%s" % (args.gcn_block, args.gcn_conv))
    result = base_line("attacked rooms/sec (%s, 4096 pts, 50 PGD iters)" % name, "rooms/s", rooms_done / elapsed, R, args, elapsed,
                       name + " dense sem_seg NB non-targeted PGD (eps=0.3, alpha=2/255, 50 iters), kNN k=16, batch=%d room(s) x "
                       "4096 pts per step (%s)" % (batch, "BASELINE configs[3]" if default_cfg else "configuration switch of configs[3]"),
                       {"steps_coalesced_per_launch": G, "device_batch_rooms": DB, "launches_in_flight_per_gpu": conc,
                        "weights": "fitted fixture (synthetic.gcn28_state_dict)" if default_cfg else "random-init",
                        "knn": "matrix (round-1 path)" if os.environ.get("PSG_GCN_KNN") == "matrix" else "fused, matrix-free"})
    result["ms_per_step"] = elapsed / (DB * n_launch / batch) * 1e3
    result["tflops_effective"] = 2 * gmac * iters * rooms_done / elapsed / 1e3 if default_cfg else None
    if R.rank == 0:
        # ---- the same attack one step (batch rooms) per launch, for readers who want the number without coalescing
        if G > 1 and not args.no_reference:
            ws1 = [runtime.GCNWorkspace(batch, NPOINT, n_blocks, **cfg) for _ in range(conc)]
            x1 = [d_images[n_warm][i * batch:(i + 1) * batch].contiguous() for i in range(G)]
            l1 = [d_labels[n_warm][i * batch:(i + 1) * batch].contiguous() for i in range(G)]
            o1 = [torch.empty_like(x) for x in x1]
            dt1 = None
            for rep in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(G):
                    launch(i, ws1, x1, l1, o1)
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
            result["uncoalesced_reference"] = {"value": batch * G / dt1, "unit": "rooms/s",
                                               "note": "one launch per step of %d room(s), %d in flight (this GPU only)" % (batch, conc)}
            del ws1
        # ---- roofline of the dominant kernel: one extra (eager) attack with HIP-event timing
        n_prof = 4
        wss[0].prof_enable(True)
        with torch.cuda.stream(streams[0]):
            wss[0].nb_attack(model, d_images[n_warm], d_labels[n_warm], 0.3, 2 / 255, n_prof, out=d_adv[n_warm])
        torch.cuda.synchronize()
        prof = wss[0].prof_read()
        wss[0].prof_enable(False)
        timed = {k: v for k, v in prof.items() if v[2] > 0}
        dom = max(timed, key=lambda k: timed[k][0])
        ms, cnt, fl = timed[dom]
        achieved = fl / (ms * 1e-3) / 1e12
        traffic, src = pmc_traffic(dom, DB, "*_pmc_traffic_gcn.json")
        result["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS,
                              "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic,
                              "traffic_source": src, "algorithmic_bytes": KNN_BYTES_PER_ROOM * DB if dom == "knn_fused" else None,
                              "avg_launch_us": ms / cnt * 1e3, "launches": cnt, "flop_per_launch": fl / cnt}
        result["kernel_ms_per_iteration"] = {k: round(v[0] / n_prof, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        if want_cpu(args, R) and default_cfg:
            from oracle import resgcn
            orc = resgcn.GCNOracle(sd, n_blocks)
            r0 = rooms[n_warm][0]
            y0 = rule_labels(r0[None])[0]
            t0 = time.time()
            resgcn.nb_step(orc, r0, r0[:, 3:6].copy(), r0[:, 3:6].copy(), y0, 2 / 255, 0.3, False)
            dt = time.time() - t0
            result["cpu_baseline"] = {"value": 1.0 / (dt * iters), "unit": "attacked rooms/s", "cores": cpu_threads(), "kind": "port",
                                      "sample": "1 room x 1 of 50 PGD iterations (28 kNN graphs, forward, backward) in %.1f s, "
                                                "extrapolated linearly" % dt}
    return result


# ======================================================================================== pointnet2_msg
def run_msg(args, R):
    """The headline attack (NB non-targeted PGD, eps=0.05, alpha=2/255, 40 iterations, batches of 8 rooms x 4096 points)
    on pointnet2_sem_seg_msg (PointNet/models/pointnet2_sem_seg_msg.py of the reference), random-init weights (no
    checkpoint of this variant ships).  BASELINE.json's metric is quoted on the SSG network."""
    import torch
    from pointsecguard_amd import runtime
    from pointsecguard_amd.synthetic import MSG_FP, MSG_SA, make_rooms, msg_state_dict, rule_labels
    sd = msg_state_dict(77)
    model = runtime.PN2Model(runtime.fold_state_dict(sd, msg=True), arch=runtime.ARCH_MSG)
    G = max(1, min(args.coalesce, args.steps))
    DB = BATCH * G
    n_groups, n_warm = -(-args.steps // G), (-(-args.warmup // G) if args.warmup > 0 else 0)
    conc = max(1, min(args.concurrency, n_groups))
    streams = [torch.cuda.Stream() for _ in range(conc)]
    wss = [runtime.PN2Workspace(DB, NPOINT, ITERS, arch=runtime.ARCH_MSG) for _ in range(conc)]
    pool, host = [], []
    for i in range(conc):
        rooms = make_rooms(DB, 9000 + 100 * R.rank + i)
        labels_h = rule_labels(rooms)
        images = torch.from_numpy(np.ascontiguousarray(rooms.transpose(0, 2, 1))).cuda()
        labels = torch.from_numpy(labels_h.astype(np.int32)).cuda()
        torch.manual_seed(100 + i)
        starts_h = torch.stack([torch.stack([torch.randint(0, n, (DB,)) for n in (NPOINT, 1024, 256, 64)])
                                for _ in range(ITERS)]).to(torch.int32)
        pool.append((images, labels, starts_h.cuda(), torch.empty_like(images)))
        host.append((rooms, labels_h, starts_h.numpy()))

    def launch(i):
        images, labels, starts, out = pool[i % conc]
        with torch.cuda.stream(streams[i % conc]):
            wss[i % conc].nb_attack(model, images, labels, starts, EPS, ALPHA, ITERS, out=out)

    for i in range(n_warm):
        launch(i)
    elapsed = R.timed(lambda: [launch(i) for i in range(n_groups)])
    rooms_done = DB * n_groups * R.world
    # algorithmic MACs per room per forward (the input-gradient pass has the same count)
    mac = 0
    for (cin, mlps), s_l in zip(MSG_SA, (1024, 256, 64, 16)):
        for mlp, k in zip(mlps, (16, 32)):
            mac += s_l * k * macs((cin + 3,) + tuple(mlp))
    for (_, cin, mlp), n_l in zip(MSG_FP, (64, 256, 1024, 4096)):
        mac += n_l * macs((cin,) + tuple(mlp))
    mac += 4096 * macs((128, 128, 13))
    result = base_line("attacked rooms/sec (PointNet++ MSG, 4096 pts, 40 PGD iters)", "rooms/s", rooms_done / elapsed, R, args,
                       elapsed, "NB non-targeted PGD (eps=0.05, alpha=2/255, 40 iters) on PointNet++ MSG sem_seg "
                       "(pointnet2_sem_seg_msg), batch=8 rooms x 4096 pts; random-init weights",
                       {"steps_coalesced_per_launch": G, "device_batch_rooms": DB, "launches_in_flight_per_gpu": conc})
    result["ms_per_step"] = elapsed / (DB * n_groups / BATCH) * 1e3
    result["tflops_effective"] = 2.0 * 2.0 * mac * ITERS * rooms_done / elapsed / 1e12
    result["gmac_per_room_forward"] = mac / 1e9
    if R.rank == 0:
        wss[0].prof_enable(True)
        launch(0)
        torch.cuda.synchronize()
        prof = wss[0].prof_read()
        wss[0].prof_enable(False)
        roof = pn2_roofline(prof, kernel_flops_msg(DB), sa_launches_per_call=2)
        fx = kernel_flops_msg_executed(DB)
        # (as on the SSG line: `frac` counts the reference's layers, `frac_executed` what the launches execute with the split
        # first layers; an SA tag's launch = its two scales, so its traffic is the two kernels' sum per call)
        roof["flop_per_launch_executed"] = fx[roof["kernel"]]
        roof["frac_executed"] = fx[roof["kernel"]] / (roof["avg_launch_us"] * 1e-6) / 1e12 / PEAK_FP32_MATRIX_TFLOPS
        roof["rooms_per_launch"] = DB
        roof["traffic"], roof["traffic_source"] = pmc_traffic(roof["kernel"], DB, "*_pmc_traffic_msg.json", PMC_SYMBOL_MSG, "sum")
        result["roofline"] = roof
        result["executed_flop_share"] = round(sum(fx[k] for k in fx) / sum(kernel_flops_msg(DB).values()), 4)
        result["kernel_ms_per_attack"] = {k: round(v[0], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        if want_cpu(args, R):
            rooms, labels_h, starts_h = host[0]
            result["cpu_baseline"] = cpu_baseline_pn2(sd, rooms[:BATCH], labels_h[:BATCH], starts_h[:, :, :BATCH],
                                                      max(1, min(args.cpu_iters, int(args.cpu_seconds / 3))), msg=True)
    return result


# ======================================================================================== randla (configs[4])
def run_randla(args, R):
    """BASELINE configs[4]: BIM colour attack (goal 'ut') on RandLA-Net with the settings the reference's tester runs
    (RandLA-Net/tester_S3DIS.py:37,142-145: NBattack = BIM with distance_metric 'l_2', magnitude 17, alpha 1.7; the l_2
    update normalises per cloud, bim.py:84-98), one 40 960-point cloud per step (ConfigS3DIS.val_batch_size = 1),
    BASELINE's 100 iterations per attack (--randla-iters) = 101 gradient steps (bim.py:204-232: one update before the
    loop), geometry (5-level k-NN pyramid) rebuilt per cloud; random-init weights (no checkpoint ships); clouds sharded
    by rank.  The l_inf variant (eps 0.05, alpha 0.01) is timed on a few launches and reported as `l_inf`.
    Network parity is UNPINNED (the reference is a TensorFlow-1 graph that cannot run here)."""
    import torch
    from pointsecguard_amd.randla import network
    from pointsecguard_amd.synthetic import randla_layer_specs, randla_params
    n_pts, iters = 40960, args.randla_iters
    params = randla_params(3)
    model = network.RandLAModel(params)
    G = max(1, min(args.rla_coalesce, args.steps))                 # clouds per launch
    n_launch, n_warm = -(-args.steps // G), (-(-args.warmup // G) if args.warmup > 0 else 0)
    conc = max(1, min(args.rla_concurrency, n_launch))
    wss = [network.RandLAWorkspace(n_pts, batch=G) for _ in range(conc)]
    streams = [torch.cuda.Stream() for _ in range(conc)]
    rng = np.random.default_rng(4 + R.rank)
    clouds, host = [], []
    for _ in range(min(n_launch + n_warm, 4)):
        xyz = (rng.random((G * n_pts, 3), dtype=np.float32) * np.array([8, 6, 3], np.float32)).astype(np.float32)
        rgb = rng.random((G * n_pts, 3), dtype=np.float32)
        lab = rng.integers(0, 13, G * n_pts)
        clouds.append((torch.from_numpy(np.concatenate([xyz, rgb], 1)).cuda(), torch.from_numpy(lab.astype(np.int32)).cuda()))
        host.append((xyz[:n_pts], rgb[:n_pts], lab[:n_pts]))

    metric = args.randla_metric
    settings = {"l_2": (17.0, 1.7), "l_inf": (0.05, 0.01)}          # (magnitude, alpha): tester_S3DIS.py:142-145 / round 1-3's line

    def step(i, n_it=iters + 1, ws_list=wss, data=clouds, metric=metric):
        f, y = data[i % len(data)]
        with torch.cuda.stream(streams[i % conc]):
            ws_list[i % conc].bim_attack(model, f, y, settings[metric][0], settings[metric][1], n_it, metric=metric)

    for i in range(n_warm):
        step(i)
    elapsed = R.timed(lambda: [step(i) for i in range(n_warm, n_warm + n_launch)])
    clouds_done = G * n_launch * R.world
    # algorithmic MACs of one forward (the input-gradient pass has about the same count minus the xyz branch)
    mac = 0
    for name, cin, cout, _ in randla_layer_specs():
        if name.startswith("Encoder_layer_"):
            lvl = int(name[len("Encoder_layer_")])
            nl = n_pts // (4 ** lvl)
            r = nl * 16 if ("LFAmlp" in name or name.endswith("fc")) else nl
        elif name == "decoder_0":
            r = n_pts // 512
        elif name.startswith("Decoder_layer_"):
            r = n_pts // (4 ** (4 - int(name[-1])))
        else:
            r = n_pts
        mac += r * cin * cout
    result = base_line("attacked clouds/sec (RandLA-Net, 40960 pts, %d BIM iters)" % iters, "clouds/s", clouds_done / elapsed,
                       R, args, elapsed, "NBattack = BIM %s colour attack (magnitude=%g, alpha=%g, %d iterations = %d gradient steps; "
                       "tester_S3DIS.py:37,142-145) on RandLA-Net, 1 cloud x 40960 pts per step (BASELINE configs[4]); random-init "
                       "weights; network parity UNPINNED (TF1 reference)" % (metric, settings[metric][0], settings[metric][1], iters, iters + 1),
                       {"steps_coalesced_per_launch": G, "launches_in_flight_per_gpu": conc, "distance_metric": metric})
    other = "l_inf" if metric == "l_2" else "l_2"
    n_o = min(n_launch, conc)                                      # a few launches of the other metric, same shape
    for i in range(n_o):
        step(i, metric=other)                                      # (its hipGraph is captured here, outside the timing)
    el_o = R.timed(lambda: [step(i, metric=other) for i in range(n_o)])
    for i in range(min(n_launch, conc)):
        step(i)                                                    # back to the headline metric's graph for what follows
    result[other] = {"value": G * n_o * R.world / el_o, "unit": "clouds/s", "note": "magnitude=%g, alpha=%g, %d launches of %d clouds"
                     % (settings[other][0], settings[other][1], n_o, G)}
    result["ms_per_step"] = elapsed / (G * n_launch) * 1e3
    result["ms_per_iteration"] = elapsed / (G * n_launch) / (iters + 1) * 1e3
    result["gmac_per_cloud_forward"] = mac / 1e9
    result["tflops_effective"] = 2.0 * 2.0 * mac * (iters + 1) * clouds_done / elapsed / 1e12
    if R.rank == 0:
        if G > 1 and not args.no_reference:
            # ---- the same attacks one cloud per launch (the reference's val_batch_size = 1)
            ws1 = [network.RandLAWorkspace(n_pts) for _ in range(conc)]
            one = [(f[i * n_pts:(i + 1) * n_pts].contiguous(), y[i * n_pts:(i + 1) * n_pts].contiguous())
                   for f, y in clouds[:1] for i in range(G)]
            dt1 = None
            for rep_ in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(len(one)):
                    step(i, iters + 1, ws1, one)
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
            result["uncoalesced_reference"] = {"value": len(one) / dt1, "unit": "clouds/s",
                                               "note": "one launch per cloud, %d in flight (this GPU only)" % conc}
            del ws1
        n_prof = 4
        wss[0].prof_enable(True)
        step(0, n_prof)
        torch.cuda.synchronize()
        kern = wss[0].prof_read_kernels()
        wss[0].prof_enable(False)
        # roofline of the kernel with the largest total time.  Every kernel of this network at these sizes is bound by
        # bandwidth / latency, not by the matrix pipe (the GEMM layers have K, M <= 512 on at most a few thousand to a few
        # hundred thousand rows): achieved = algorithmic bytes of its launches / their HIP-event time against the HBM
        # peak; the FLOP-based fraction of the same launches rides along, and `traffic` is the counted HBM volume per
        # launch from the committed PMC passes of this command (profiles/*_pmc_traffic_randla.json) when they exist
        dom = max(kern, key=lambda k: kern[k][0])
        ms, cnt, fl, by = kern[dom]
        anon = "(anonymous namespace)::"
        symbol = {"gemm_rows_kernel<2,2,.,.,1,1> (64x64 tiles)": ("void psg::gemm_rows_kernel<2, 2, 0, false, 1, 1>(psg::GemmArgs)",
                                                                 "void psg::gemm_rows_kernel<2, 2, 3, false, 1, 1>(psg::GemmArgs)"),
                  "gemm_rows_kernel<2,2> (128x128 tiles)": ("void psg::gemm_rows_kernel<2, 2, 0, false, 2, 2>(psg::GemmArgs)",
                                                            "void psg::gemm_rows_kernel<2, 2, 3, false, 2, 2>(psg::GemmArgs)"),
                  "direct_gemm_kernel<.,2> (33-64 channels, levels 0-1)": ("void (anonymous namespace)::direct_gemm_kernel<0, 2>(psg::GemmArgs)",
                                                                          "void (anonymous namespace)::direct_gemm_kernel<3, 2>(psg::GemmArgs)"),
                  "direct_gemm_kernel<.,1> / skinny (<= 32 channels)": ("void (anonymous namespace)::direct_gemm_kernel<0, 1>(psg::GemmArgs)",
                                                                        "void (anonymous namespace)::direct_gemm_kernel<3, 1>(psg::GemmArgs)",
                                                                        "void (anonymous namespace)::skinny_gemm_kernel<0>(psg::GemmArgs)",
                                                                        "void (anonymous namespace)::skinny_gemm_kernel<3>(psg::GemmArgs)")}.get(dom, (anon + dom,))
        traffic, src = None, None
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_randla.json")))
        if files:
            with open(files[-1]) as fh:
                table = json.load(fh)
            # (the counter summaries key a kernel by its full symbol, arguments included)
            rows_ = [v for k, v in table.items() if k != "_meta" and any(k == s_ or k.startswith(s_ + "(") for s_ in symbol)]
            if rows_ and table.get("_meta", {}).get("device_batch_rooms") == G:
                n_l = sum(r["launches_fetch_pass"] for r in rows_)
                traffic = sum(r["hbm_bytes_per_launch"] * r["launches_fetch_pass"] for r in rows_) / max(n_l, 1)
                src = "profiles/" + os.path.basename(files[-1])
        gbs = by / (ms * 1e-3) / 1e9
        is_gemm = dom in network.RandLAWorkspace.PROF_KERNELS[:4]
        mfma_frac = fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS if is_gemm else 0.0
        result["roofline"] = {"bound": "hbm", "kernel": dom + (" (linear and leaky-ReLU epilogues" if is_gemm else " (levels 1-4: attention "
                              "scores T[neigh] + S2, softmax over the 16 neighbours, pooling" if "split" in dom else " (level 0: scores, "
                              "softmax and pooling of a 16-channel attentive pooling in one kernel") + "; %d cloud(s) per launch)" % G,
                              "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                              "traffic": traffic, "traffic_source": src, "algorithmic_bytes": by / cnt,
                              "avg_launch_us": ms / cnt * 1e3, "launches": cnt,
                              "share_of_profiled_time": ms / sum(v[0] for v in kern.values())}
        if is_gemm:
            result["roofline"]["mfma_frac_same_launches"] = mfma_frac
            if mfma_frac > gbs / PEAK_HBM_GBS:
                # (round 5: with the narrow layers and the poolings faster, the family with the largest time can be the wide GEMMs of
                # the coarse levels, which the matrix pipe binds, not the memory: the line then prices it against that peak and keeps
                # the bytes-based figure beside it)
                result["roofline"].update({"bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": PEAK_FP32_MATRIX_TFLOPS,
                                           "unit": "TFLOP/s", "frac": mfma_frac, "hbm_frac_same_launches": gbs / PEAK_HBM_GBS,
                                           "flop_per_launch": fl / cnt})
        result["kernel_family_gbs"] = {k: round(v[3] / (v[0] * 1e-3) / 1e9, 1) for k, v in kern.items()}
        result["profiled_kernels_ms_per_iteration"] = {k: round(v[0] / n_prof, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0])}
        if want_cpu(args, R):
            from oracle import randla, randla_net
            xyz, rgb, lab = host[0]
            t0 = time.time()
            use_ref = randla.have_ref()
            pts, neigh, pools, ups = randla.pyramid(xyz[None], knn=randla.knn_ref if use_ref else randla.knn_brute)
            pyr = ([p[0] for p in pts], [q[0] for q in neigh], [p[0] for p in pools], [u[0] for u in ups])
            t_pyr = time.time() - t0
            orc = randla_net.RandLAOracle(params)
            t0 = time.time()
            _, _, grad = randla_net.loss_and_grad(orc, xyz, rgb, lab, pyr)
            randla_net.bim_step(rgb.reshape(-1), rgb.reshape(-1), grad.reshape(-1), 0.05, 0.01)     # (either metric: noise next to the gradient)
            t_it = time.time() - t0
            result["cpu_baseline"] = {"value": 1.0 / (t_pyr + (iters + 1) * t_it), "unit": "attacked clouds/s", "cores": cpu_threads(),
                                      "kind": "port", "sample": "1 cloud: index pyramid (%s) %.1f s + 1 of %d BIM iterations %.1f s, "
                                      "iterations extrapolated linearly" % ("the reference's knn_.cxx" if use_ref else "numpy", t_pyr,
                                                                            iters + 1, t_it)}
    return result


# ======================================================================================== tarnu (configs[2])
def run_tarnu(args, R):
    """BASELINE configs[2]: targeted NU attack (Adam in tanh space on the colour channels of a masked object class) on
    PointNet++ sem_seg, 32 rooms per step, through the public API with the harness values of NU_target_test_semseg.py:181
    (c = 1, kappa = 0, lr = 0.01, target = 6).

    --nu-mode per-room (default): the attack is applied PER ROOM - the reference's semantics are
    batch-of-one (target.py:62-133: losses, Smooth term and the early-exit accuracies all read batch row 0) - with a 40-step
    cap for timing; every room runs until its own `target_acc > 0.9` exit or the cap.  A step = 32 rooms (--scaling strong:
    32 / n_gpus rooms per rank, the "sharded 8x" of BASELINE configs[2]).
    --nu-mode batch32: one call on a [32, 9, 4096] batch.  The reference's exit test then counts the masked points of all
    32 rows against ONE row's mask count (target.py:105-121), so it fires after the first optimiser step: that quirk is
    kept, and the line is reported for completeness (`batch32_quirk` of the default run)."""
    result = tarnu_measure(args, R, args.nu_mode)
    if args.nu_mode == "per-room" and R.rank == 0 and R.world == 1 and not args.no_reference:
        import copy
        a1 = copy.copy(args)
        a1.steps, a1.warmup, a1.no_cpu_baseline = 3, 1, True
        q = tarnu_measure(a1, R, "per-room-calls", with_roofline=False)
        result["uncoalesced_reference"] = {"value": q["value"], "unit": "rooms/s",
                                           "note": "one call per room (batches of one), %d calls in flight" % q["config"]["attacks_in_flight"]}
        import copy
        a2 = copy.copy(args)
        a2.steps, a2.warmup, a2.no_cpu_baseline = 3, 1, True
        q = tarnu_measure(a2, R, "batch32", with_roofline=False)
        result["batch32_quirk"] = {k: q[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "optimizer_steps_per_attack",
                                                     "room_steps_per_sec") if k in q}
        result["batch32_quirk"]["note"] = ("one call on 32 rooms: the reference's exit test compares the target hits of all 32 rows "
                                           "with one row's mask count (target.py:105-121) and stops after the first optimiser step")
    return result


def tarnu_measure(args, R, mode, with_roofline=True):
    import threading
    from concurrent.futures import ThreadPoolExecutor

    import torch
    from pointsecguard_amd.attacks import torchattacks
    from pointsecguard_amd.attacks.torchattacks.attacks import nu as nu_mod
    from pointsecguard_amd.models.pointnet2_sem_seg import get_model
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    target, src_cls = 6, 2
    if getattr(args, "nu_plan_ahead", 0):
        nu_mod.plan_ahead = int(args.nu_plan_ahead)
    strong = args.scaling == "strong"
    per_step = max(1, 32 // R.world) if strong else 32          # rooms of one step on this rank
    cap = args.nu_steps if args.nu_steps else (40 if mode in ("per-room", "per-room-calls") else 100)
    sd = dict(np.load(os.path.join(ROOT, "tests", "golden", "pn2_weights.npz")))
    n_steps = args.steps + args.warmup
    # jobs: (step, first room, rooms) - one attack call each
    lockstep = mode == "per-room" and per_step >= 2          # one call advances the rooms of a step in lockstep (forward_rooms)
    per_room = mode in ("per-room", "per-room-calls")
    # lockstep: --nu-coalesce consecutive steps (independent rooms) share one forward_rooms call, dealt evenly like the
    # headline workload's launches; the other modes keep one group per step
    coal = max(1, args.nu_coalesce) if lockstep else 1
    def split(n):
        n_l = -(-n // coal) if n else 0
        return [n // n_l + (1 if i < n % n_l else 0) for i in range(n_l)]
    g_warm, g_timed = split(args.warmup), split(args.steps)
    g_sizes, n_warm_g = g_warm + g_timed, len(g_warm)
    rooms, s0 = [], 0
    for gs in g_sizes:
        rooms.append(np.concatenate([make_rooms(per_step, 7000 + 1000 * R.rank + s0 + i, structured=True) for i in range(gs)]))
        s0 += gs
    labels = [rule_labels(r) for r in rooms]
    def jobs_of(g):
        return [(g, b, 1) for b in range(len(rooms[g]))] if per_room and not lockstep else [(g, 0, len(rooms[g]))]
    n_jobs_timed = sum(len(jobs_of(g)) for g in range(n_warm_g, len(g_sizes)))
    lock_conc = int(os.environ.get("PSG_BENCH_NU_LOCK_CONC", "4"))          # (3 / 4 / 5 calls in flight: 597 / 607 / 594 rooms/s, tools/r05_o.sh)
    conc = max(1, min(args.nu_concurrency if not lockstep else min(args.nu_concurrency, lock_conc), n_jobs_timed))
    nets, streams = [], [torch.cuda.Stream() for _ in range(conc)]
    for _ in range(conc):
        net = get_model(13)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(net.cuda().eval())
    for net in nets:
        # the device copy of every model instance is built HERE, before any thread starts: built lazily by its first attack
        # it meant allocations and uploads beside another thread's hipGraph capture (one run in three lost a capture to it)
        net._packed()
    d_images = [torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 1))).cuda() for r in rooms]
    opt_steps, exits, lock = [0], [0], threading.Lock()

    def attack(job, slot):
        step, b0, nb = job
        mask = labels[step][b0] == src_cls                      # the harness masks by the batch's first room (mask[0])
        with torch.cuda.stream(streams[slot]):
            atk = torchattacks.tar_NU_attack(nets[slot], c=1, kappa=0, steps=cap, lr=0.01, target=target, mask=mask)
            if lockstep:
                masks = labels[step][b0:b0 + nb] == src_cls     # every room masks its own points of the source class
                out, steps_run = atk.forward_rooms(d_images[step][b0:b0 + nb], labels[step][b0:b0 + nb].astype(np.float64), masks)
            else:
                out, n_run = nu_mod.nu_attack(atk, d_images[step][b0:b0 + nb], labels[step][b0:b0 + nb].astype(np.float64), mask, target,
                                              5, targeted_variant=True, return_steps=True)
                steps_run = np.array([n_run])
            streams[slot].synchronize()
        with lock:
            opt_steps[0] += int(steps_run.sum())
            exits[0] += int((steps_run < cap).sum())
        return out

    def run(lo, hi):
        # slot = i % conc: attacks of one slot run in order on that slot's thread, so a model instance is never shared
        jobs = [j for s in range(lo, hi) for j in jobs_of(s)]

        def worker(slot):
            for i in range(slot, len(jobs), conc):
                attack(jobs[i], slot)
        if conc == 1:
            worker(0)
            return
        with ThreadPoolExecutor(max_workers=conc) as pool:
            list(pool.map(worker, range(conc)))

    torch.manual_seed(R.rank)
    run(0, n_warm_g)
    torch.cuda.synchronize()
    opt_steps[0], exits[0] = 0, 0
    elapsed = R.timed(lambda: run(n_warm_g, len(g_sizes)))
    total_opt, total_exit = R.sum(opt_steps[0]), R.sum(exits[0])
    n_attacks = (per_step if per_room else 1) * args.steps * R.world
    rooms_per_attack = 1 if per_room else per_step
    rooms_per_call = len(rooms[n_warm_g]) if lockstep or not per_room else 1
    result = base_line("attacked rooms/sec (tar_NU, 4096 pts, <= %d Adam steps)" % cap, "rooms/s",
                       per_step * args.steps * R.world / elapsed, R, args, elapsed,
                       "tar_NU_attack (c=1, kappa=0, lr=0.01, target=6, neighbour=5) on PointNet++ SSG sem_seg, %s (BASELINE "
                       "configs[2]); fitted fixture weights"
                       % ("applied per room, %d rooms x 4096 pts per step and GPU%s" % (per_step, ", the rooms of a step advanced in "
                          "lockstep (forward_rooms: one launch per operation), %d step(s) per call" % coal if lockstep else ", one call per room") if per_room
                          else "one call on a batch of %d rooms x 4096 pts" % per_step),
                       {"mode": mode, "rooms_per_step_per_gpu": per_step, "optimizer_steps_cap": cap,
                        ("calls_in_flight" if lockstep else "attacks_in_flight"): conc, "rooms_per_call": rooms_per_call})
    result["scaling"] = "strong" if strong else "weak"
    result.update({"optimizer_steps_per_sec": total_opt / elapsed, "optimizer_steps_run": int(total_opt),
                   "optimizer_steps_per_attack": total_opt / n_attacks, "attacks_that_reached_the_target": int(total_exit),
                   "attacks": int(n_attacks), "room_steps_per_sec": rooms_per_attack * total_opt / elapsed})
    if R.rank == 0 and with_roofline:
        # roofline: the network kernels of one more attack, HIP events on its launch stream
        ws = nets[0]._workspace(rooms_per_call, NPOINT, nu_mod.CHUNK + 1)
        ws.prof_enable(True)
        attack(jobs_of(n_warm_g)[0], 0)
        torch.cuda.synchronize()
        prof = ws.prof_read()
        ws.prof_enable(False)
        result["roofline"] = pn2_roofline(prof, kernel_flops(rooms_per_call))
        result["roofline"]["rooms_per_launch"] = rooms_per_call
        result["roofline"]["traffic"], result["roofline"]["traffic_source"] = pmc_traffic(result["roofline"]["kernel"], rooms_per_call,
                                                                                          "*_pmc_traffic_tarnu.json")
        result["kernel_ms_per_attack"] = {k: round(v[0], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        if want_cpu(args, R):
            from oracle import attacks as oatk
            from oracle import pn2
            orc = pn2.PN2Oracle(sd)
            bs = 4
            img = np.ascontiguousarray(rooms[0][:bs].transpose(0, 2, 1))
            lab = labels[0][:bs]
            mask = lab[0] == src_cls
            w = oatk.inverse_tanh_space(img[:, 3:6][:, :, mask])
            m, v = np.zeros_like(w), np.zeros_like(w)
            rng = np.random.default_rng(0)
            starts = np.stack([rng.integers(0, n, (bs,)) for n in (NPOINT, 1024, 256, 64)]).astype(np.int32)
            t0, n_done = time.time(), 0
            while n_done < 1 or (time.time() - t0 < args.cpu_seconds / 2 and n_done < 4):
                r = oatk.nu_step(orc, img, img.copy(), w, m, v, n_done + 1, lab, starts, 1.0, 0.0, 0.01, 5, mask=mask, target=target)
                w, m, v = r["w"], r["m"], r["v"]
                n_done += 1
            dt = time.time() - t0
            result["cpu_baseline"] = {"value": bs * n_done / dt, "unit": "optimiser room-steps/s (compare room_steps_per_sec)",
                                      "cores": cpu_threads(), "kind": "port",
                                      "sample": "%d rooms x %d optimiser steps (forward, f / Smooth / L2 losses, backward, Adam) in "
                                                "%.1f s" % (bs, n_done, dt)}
    return result


if __name__ == "__main__":
    main()
