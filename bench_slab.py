"""bench_slab.py — the N > 1 leg of bench.py: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI), x-slab decomposition with one halo exchange per step (pi-sph-fluid_amd/slab.py, include/sph.h).

Weak scaling: the cfg2 -> cfg3 family, one 4000 x 500 lattice block (2 000 000 particles, 1200 m of box) per GPU;
`value` is the whole-job Mparticle-steps/s = N_fluid_total x timesteps/s / 1e6, timed over exactly K steps
between barrier + device synchronisation, MAX over ranks.  SPH_SLAB_TRANSPORT=host selects a host-staged gloo
transport (all ranks may then share one device; used to exercise this file on a 1-GPU box)."""
import importlib
import json
import os
import sys
import time

import numpy as np

HBM_PEAK_GBS = 8000.0


def log(rank, *a):
    print("[rank %d]" % rank, *a, file=sys.stderr, flush=True)


class HostStagedTransport:
    """gloo + CPU tensors, halo buffers staged through the host (sph_slab_copy_out / _copy_in)."""

    def __init__(self, torch, dist, slab, rank, world):
        self.torch, self.dist, self.slab, self.rank, self.world = torch, dist, slab, rank, world
        self.rl = torch.zeros(slab.words, dtype=torch.int32)
        self.rr = torch.zeros(slab.words, dtype=torch.int32)

    def reduce_flag(self):
        if self.world > 1:
            f = self.torch.tensor([self.slab.flag_get()], dtype=self.torch.int32)
            self.dist.all_reduce(f, op=self.dist.ReduceOp.MAX)
            self.slab.flag_set(int(f.item()))

    def exchange_start(self):
        self.exchange()

    def exchange_finish(self, handle):
        pass

    def exchange(self):
        t, d, ops = self.torch, self.dist, []
        if self.rank > 0:
            sl = t.from_numpy(self.slab.copy_out(0).view(np.int32))
            ops += [d.P2POp(d.isend, sl, self.rank - 1), d.P2POp(d.irecv, self.rl, self.rank - 1)]
        if self.rank < self.world - 1:
            sr = t.from_numpy(self.slab.copy_out(1).view(np.int32))
            ops += [d.P2POp(d.isend, sr, self.rank + 1), d.P2POp(d.irecv, self.rr, self.rank + 1)]
        for w in (d.batch_isend_irecv(ops) if ops else []):
            w.wait()
        if self.rank > 0:
            self.slab.copy_in(0, self.rl.numpy().view(np.uint32))
        if self.rank < self.world - 1:
            self.slab.copy_in(1, self.rr.numpy().view(np.uint32))


def run_slabs(sph, args, emit):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    host_staged = os.environ.get("SPH_SLAB_TRANSPORT", "") == "host"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    ndev = torch.cuda.device_count()
    device = local_rank % max(ndev, 1)
    torch.cuda.set_device(device)
    dist.init_process_group("gloo" if host_staged else "nccl", rank=rank, world_size=world)

    prm, f, b = sph.dam_break(world)
    n_total = len(f)
    parts = sph.slab.partition_columns(prm, f, world)
    c0, c1 = parts[rank]
    t0 = time.time()
    slab = sph.slab.GpuSlab(sph, prm, f, b, c0, c1, rank > 0, rank < world - 1, 0.0, -9.81, device=device)
    del f
    stream = torch.cuda.Stream(device=device)
    gx, gy = 0.0, -9.81
    with torch.cuda.stream(stream):
        slab.set_stream(stream.cuda_stream)          # kernels and RCCL ops are ordered through torch's current stream
        transport = (HostStagedTransport(torch, dist, slab, rank, world) if host_staged
                     else sph.slab.TorchTransport(torch, dist, slab, rank, world, torch.device("cuda", device)))
        runner = sph.slab.SlabRunner(slab, transport)
        log(rank, "slab columns [%d,%d) created in %.2fs, local/owned = %s" % (c0, c1, time.time() - t0, slab.counts()))
        runner.step(args.warmup, gx, gy)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.step(args.steps, gx, gy)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        slab.sync()                                   # raises on capacity / out-of-domain / NaN
        n_loc, n_own = slab.counts()
    tmax = torch.tensor([dt], dtype=torch.float64)
    owned = torch.tensor([n_own], dtype=torch.int64)
    if not host_staged:
        tmax, owned = tmax.cuda(), owned.cuda()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(owned, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    if rank == 0:
        steps_per_s = args.steps / dt
        value = steps_per_s * n_total / 1e6
        out = {
            "metric": "SPH Mparticle-steps/sec (N_fluid x timesteps/sec / 1e6)",
            "value": round(value, 2), "unit": "Mparticle-steps/s", "timesteps_per_s": round(steps_per_s, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "dam break, %d x-slabs of 2 000 000 fluid particles (cfg2 family; N = 4 is cfg3): "
                                   "%d fluid + %d boundary particles, box %d x 60 m" % (world, n_total, len(b), 1200 * world),
                       "n_fluid": n_total, "n_boundary": len(b),
                       "parallelism": "%d x-slabs, 2-column halo + migration in one RCCL send/recv pair per neighbour per step%s"
                                      % (world, " (host-staged gloo transport)" if host_staged else "")},
            "particles_conserved": int(owned.item()) == n_total,
            "roofline": {"bound": "hbm", "achieved": round(sph.STEP_ALGO_BYTES * n_total * steps_per_s / 1e9 / world, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU (whole step, 152 B per particle-step)",
                         "frac": round(sph.STEP_ALGO_BYTES * n_total * steps_per_s / 1e9 / world / HBM_PEAK_GBS, 4),
                         "traffic": None},
        }
        emit(out)
    dist.barrier()
    dist.destroy_process_group()
    slab.close()
