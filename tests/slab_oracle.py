"""TEST INFRASTRUCTURE: a CPU slab backend built on the oracle, with the same interface as the product's GpuSlab
(step_begin / step_pack / step_end / halo_tensors / flag_tensor / read), so that the product's host-side slab logic (partitioner, halo
protocol, TorchTransport, SlabRunner) can be exercised by world_size-2 gloo tests without a GPU.

The backend keeps its local arrays ordered by global particle id, so the oracle's neighbour summation order
(ascending index inside a cell) is the same as in a single-rank run: results must match BIT-EXACTLY, which
makes any missing ghost, lost migrant or duplicated particle visible."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


class OracleSlab:
    def __init__(self, sph, orc, O, prm, fluid, boundary_psi, c0, c1, has_left, has_right, gx, gy, halo_capacity, local=None):
        self.sph, self.orc, self.O, self.prm = sph, orc, O, prm
        self.p = O.params((prm.x_min, prm.x_max, prm.y_min, prm.y_max))
        self.b = boundary_psi
        self.c0, self.c1, self.has_left, self.has_right = c0, c1, has_left, has_right
        self.cap = halo_capacity
        words = sph.slab.halo_words(halo_capacity)
        self.bufs = [np.zeros(words, np.uint32) for _ in range(4)]     # send_l, send_r, recv_l, recv_r
        self.step = 0                                                   # steps so far: full records carry the step they are for (include/sph.h)
        self.half_dt = 0.5 * float(np.float32(prm.dt))                  # 0.5*DT in double (:616)
        self.dt = np.float32(prm.dt)
        loc, ids = local if local is not None else sph.slab.local_subset(prm, fluid, c0, c1)
        self._evaluate(loc.view(orc.PARTICLE).copy(), ids, gx, gy, second_kick=False)

    def close(self):
        pass

    def _evaluate(self, loc, ids, gx, gy, second_kick):
        order = np.argsort(ids, kind="stable")
        loc, ids = np.ascontiguousarray(loc[order]), ids[order]
        assert len(np.unique(ids)) == len(ids), "duplicate particle after exchange"
        du, dv = self.O.eval(self.p, loc, self.b, gx, gy, threads=2)
        gc = self.sph.slab.global_columns(self.prm, loc["x"])
        own = (gc >= self.c0) & (gc < self.c1)
        loc, ids, du, dv = loc[own], ids[own], du[own], dv[own]
        if second_kick:                                                  # :637-640
            loc["u"] = (loc["u"].astype(np.float64) + self.half_dt * du.astype(np.float64)).astype(np.float32)
            loc["v"] = (loc["v"].astype(np.float64) + self.half_dt * dv.astype(np.float64)).astype(np.float32)
        self.own, self.ids, self.du, self.dv = loc, ids, du, dv

    def _pack(self, buf, sel):
        n = int(sel.sum())
        assert n <= self.cap, "halo capacity"
        buf[:] = 0
        buf[2], buf[3] = n, self.step      # header {update count, its step, record count, their step}: this backend sends records every step
        rec = np.zeros((n, 5), np.uint32)
        for k, f in enumerate(("x", "y", "u", "v")):
            rec[:, k] = self.own[f][sel].view(np.uint32)
        rec[:, 4] = self.ids[sel]
        buf[4:4 + 5 * n] = rec.ravel()

    def flag_tensor(self, torch, device):
        """the rebuild word: this backend rebuilds (and sends full records) every step, like the reference (:626)."""
        if getattr(self, "_flag", None) is None:
            self._flag = torch.zeros(1, dtype=torch.int32)
        return self._flag

    def step_begin(self, gx, gy):
        self.g = (gx, gy)
        self.step += 1
        if getattr(self, "_flag", None) is not None:
            self._flag[0] = 1
        o = self.own
        o["u"] = (o["u"].astype(np.float64) + self.half_dt * self.du.astype(np.float64)).astype(np.float32)   # :616
        o["v"] = (o["v"].astype(np.float64) + self.half_dt * self.dv.astype(np.float64)).astype(np.float32)
        o["x"] = o["x"] + self.dt * o["u"]                                                                     # :622
        o["y"] = o["y"] + self.dt * o["v"]

    def step_overlap(self):
        pass

    def step_pack(self):
        if getattr(self, "_flag", None) is not None:
            assert int(self._flag[0]) == 1            # MAX over ranks of words that are all 1
        o = self.own
        gc = self.sph.slab.global_columns(self.prm, o["x"])
        self._pack(self.bufs[0], (gc < self.c0 + 2) if self.has_left else np.zeros(len(o), bool))
        self._pack(self.bufs[1], (gc >= self.c1 - 2) if self.has_right else np.zeros(len(o), bool))

    def _unpack(self, buf):
        n = int(buf[2])
        assert int(buf[3]) == self.step, "records of another step: ranks out of step"
        rec = buf[4:4 + 5 * n].reshape(n, 5)
        q = np.zeros(n, self.orc.PARTICLE)
        for k, f in enumerate(("x", "y", "u", "v")):
            q[f] = rec[:, k].copy().view(np.float32)
        q["m"] = self.own["m"][0] if len(self.own) else 0
        q["rho"] = 1000.0
        return q, rec[:, 4].astype(np.uint32)

    def step_end(self):
        parts, ids = [self.own], [self.ids]
        if self.has_left:
            q, i = self._unpack(self.bufs[2]); parts.append(q); ids.append(i)
        if self.has_right:
            q, i = self._unpack(self.bufs[3]); parts.append(q); ids.append(i)
        self._evaluate(np.concatenate(parts), np.concatenate(ids), self.g[0], self.g[1], second_kick=True)

    def halo_tensors(self, torch, device):
        return [torch.from_numpy(b.view(np.int32)) for b in self.bufs]

    def read(self):
        return self.own, self.ids, self.du, self.dv


def gloo_worker(rank, world, port, scene, nsteps, q, rebalance_at=0):
    """one gloo rank: build my slab, run the product's SlabRunner over TorchTransport, report my particles."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sph = importlib.import_module("pi-sph-fluid_amd")
        import orc
        O = orc.Oracle("strict")
        prm, f, b = scene_build(sph, scene)
        bp = b.view(orc.PARTICLE).copy()
        O.psi(O.params((prm.x_min, prm.x_max, prm.y_min, prm.y_max)), bp)
        parts = sph.slab.partition_columns(prm, f, world, slack=8)
        c0, c1 = parts[rank]
        slab = OracleSlab(sph, orc, O, prm, f, bp, c0, c1, rank > 0, rank < world - 1, 0.0, -9.81,
                          sph.slab.default_halo_capacity(prm))
        cap = sph.slab.default_halo_capacity(prm)

        def factory(a, z, hl, hr, loc, ids, gx, gy):
            return OracleSlab(sph, orc, O, prm, None, bp, a, z, hl, hr, gx, gy, cap, local=(loc, ids))

        runner = sph.slab.SlabRunner(slab, sph.slab.TorchTransport(torch, dist, slab, rank, world, "cpu"),
                                     factory=factory, prm=prm, rank0=rank, world=world)
        migrated = 0
        before = set(slab.ids.tolist())
        new_parts = None
        if rebalance_at:
            runner.step(rebalance_at, 0.0, -9.81)
            new_parts = runner.rebalance(0.0, -9.81, min_gain=0.0)
            runner.step(nsteps - rebalance_at, 0.0, -9.81)
        else:
            runner.step(nsteps, 0.0, -9.81)
        slab = runner.slabs[0]
        if new_parts:
            c0, c1 = new_parts[rank]
        own, ids, du, dv = slab.read()
        migrated = len(set(ids.tolist()) - before)
        q.put((rank, own, ids, du, dv, migrated, (c0, c1)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def scene_build(sph, scene):
    if scene == "block_moving":
        # a block thrown sideways at 30 m/s so that particles cross the slab boundary within a few steps
        prm, f, b = sph.scene_block((0.0, 12.0, 0.0, 4.0), 2.0, 0.4, 60, 20)
        f["u"] = 30.0
        f["v"] = np.linspace(-3, 3, len(f)).astype(np.float32)
        return prm, f, b
    if scene == "drop":
        return sph.scene("cfg0")
    raise ValueError(scene)
