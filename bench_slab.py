"""bench_slab.py — the N > 1 leg of bench.py: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI), x-slab decomposition with one halo exchange per step (pi-sph-fluid_amd/slab.py, include/sph.h).

Workloads
  cfg2 (default)  weak scaling, the cfg2 -> cfg3 family: one 4000 x 500 lattice block (2 000 000 particles, 1200 m of
                  box) per GPU; N = 4 is cfg3.  Constant gravity.
  cfg3            the 8M dam break as BASELINE.json states it (any N; 4 slabs is the named configuration).
  cfg4            32 000 000 particles, box 2400.6 x 150 m, gravity from the scripted tilt trace (sph_gravity: g(t) =
                  G (sin th, -cos th), th = 15 deg sin(2 pi t / 8 s), re-sampled every 0.1 s of simulated time like the
                  reference's 10 Hz poll, pi_sph_fluid.c:455-461).  Fixed total size: strong scaling.
`value` is the whole-job Mparticle-steps/s = N_fluid_total x timesteps/s / 1e6, timed over exactly K steps between
barrier + device synchronisation, MAX over ranks.  Every rank generates only the lattice columns it holds.
--transport host selects a host-staged gloo transport (all ranks may then share one device; used to exercise this
file on a 1-GPU box)."""
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
        self.torch, self.dist, self.slab, self.rank, self.world, self.device = torch, dist, slab, rank, world, "cpu"
        self.rl = torch.zeros(slab.words, dtype=torch.int32)
        self.rr = torch.zeros(slab.words, dtype=torch.int32)

    def rebind(self, slabs):          # re-balancing: a new slab context took the place of the old one
        self.slab = slabs[0]

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


def _borrow_collectives(sph):
    """the re-balancing collectives of TorchTransport work on CPU tensors over gloo as they are"""
    T = sph.slab.TorchTransport
    HostStagedTransport._dev = T._dev
    HostStagedTransport.sum_over_ranks = T.sum_over_ranks
    HostStagedTransport.redistribute = T.redistribute


def workload_spec(sph, name, world):
    """(label, block spec, scaling, gravity source or None)"""
    if name in ("cfg2", "dam"):
        spec = sph.dam_break_spec(world)
        return ("dam break, %d x-slabs of 2 000 000 fluid particles (cfg2 family; N = 4 is cfg3)" % world, spec, "weak", None)
    if name == "cfg3":
        return ("cfg3: 8M dam break", sph.BLOCK_SCENES["cfg3"], "strong", None)
    if name == "cfg4":
        grav = sph.GravitySource(sph.GRAVITY_TILT, 9.81)      # 15 deg, 8 s, 0.1 s hold: the defaults of sph_gravity_init
        return ("cfg4: 32M tank under the scripted tilt trace", sph.BLOCK_SCENES["cfg4"], "strong", grav)
    raise ValueError("no slab workload %r" % name)


def run_slabs(sph, args, emit):
    import torch
    import torch.distributed as dist

    _borrow_collectives(sph)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    host_staged = args.transport == "host"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    ndev = torch.cuda.device_count()
    device = local_rank % max(ndev, 1)
    if not host_staged and world > ndev:
        raise SystemExit("bench.py: %d ranks need %d GPUs for the RCCL transport (found %d); use --transport host to "
                         "rehearse on fewer" % (world, world, ndev))
    if ndev:
        torch.cuda.set_device(device)
    dist.init_process_group("gloo" if host_staged else "nccl", rank=rank, world_size=world)

    label, spec, scaling, grav = workload_spec(sph, args.workload, world)
    box, x0, y0, nx, ny = spec
    prm = sph.default_params(box)
    n_total = nx * ny
    walls = sph.scene_walls(prm)
    parts = sph.slab.partition_block(prm, spec, world)
    c0, c1 = parts[rank]
    t0 = time.time()
    loc, ids = sph.slab.local_block_subset(sph, prm, spec, c0, c1)      # this rank's columns (+ 2 ghost columns) only
    g0 = grav.sample(0.0) if grav else (0.0, -9.81)
    slab = sph.slab.GpuSlab(sph, prm, None, walls, c0, c1, rank > 0, rank < world - 1, g0[0], g0[1], device=device,
                            local=(loc, ids))
    del loc, ids
    one_launch = not host_staged      # one rank per GPU: nothing else computes on the device (sph_set_rebuild_launches)
    slab.set_rebuild_launches(one_launch)
    dt_sim = float(np.float32(prm.dt))
    stream = torch.cuda.Stream(device=device)
    sim_step = [0]

    def gravity(_k):
        """the gravity vector of the next step (:632 re-reads g every step)"""
        sim_step[0] += 1
        return grav.sample(sim_step[0] * dt_sim) if grav else (0.0, -9.81)

    with torch.cuda.stream(stream):
        slab.set_stream(stream.cuda_stream)          # kernels and RCCL ops are ordered through torch's current stream
        transport = (HostStagedTransport(torch, dist, slab, rank, world) if host_staged
                     else sph.slab.TorchTransport(torch, dist, slab, rank, world, torch.device("cuda", device)))
        def factory(a, z, hl, hr, loc_, ids_, gx_, gy_):
            new = sph.slab.GpuSlab(sph, prm, None, walls, a, z, hl, hr, gx_, gy_, device=device, local=(loc_, ids_))
            new.set_rebuild_launches(one_launch)
            return new

        runner = sph.slab.SlabRunner(slab, transport, factory=factory, prm=prm, rank0=rank, world=world)
        log(rank, "slab columns [%d,%d) of %d created in %.2fs, local/owned = %s" %
            (c0, c1, sph.slab.grid_columns(prm), time.time() - t0, slab.counts()))
        runner.step(args.warmup, gravity=gravity)
        if getattr(args, "rebalance", False):
            # dynamic re-balancing, once, outside the timed region: new column ranges from the current
            # per-column histogram; the re-created slab adopts the stream again
            g_now = grav.sample(sim_step[0] * dt_sim) if grav else (0.0, -9.81)
            new_parts = runner.rebalance(g_now[0], g_now[1], min_gain=0.0)
            runner.slabs[0].set_stream(stream.cuda_stream)
            slab = runner.slabs[0]
            log(rank, "re-balanced: columns", new_parts[rank] if new_parts else "unchanged", "local/owned =", slab.counts())
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.step(args.steps, gravity=gravity)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        slab.sync()                                   # raises on capacity / out-of-domain / NaN
        n_loc, n_own = slab.counts()
        rebuilds = slab.rebuilds()
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
        step_gbs = sph.STEP_ALGO_BYTES * n_total * steps_per_s / 1e9 / world
        out = {
            "metric": "SPH Mparticle-steps/sec (N_fluid x timesteps/sec / 1e6)",
            "value": round(value, 2), "unit": "Mparticle-steps/s", "timesteps_per_s": round(steps_per_s, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d fluid + %d boundary particles, box %g x %g m%s" %
                                   (label, n_total, len(walls), box[1] - box[0], box[3] - box[2],
                                    ", gravity = scripted tilt trace (15 deg, 8 s period, 0.1 s hold)" if grav else ""),
                       "n_fluid": n_total, "n_boundary": len(walls),
                       "parallelism": "%d x-slabs, 2-column halo + migration in one RCCL send/recv pair per neighbour per step%s"
                                      % (world, " (host-staged gloo transport)" if host_staged else "")},
            "particles_conserved": int(owned.item()) == n_total,
            "neighbour_rebuilds_per_step": round(rebuilds / max(args.steps + args.warmup, 1), 4),
            "roofline": {"bound": "hbm", "achieved": round(step_gbs, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU (whole step, 152 B per particle-step)",
                         "frac": round(step_gbs / HBM_PEAK_GBS, 4), "traffic": None},
        }
        emit(out)
    dist.barrier()
    dist.destroy_process_group()
    slab.close()
