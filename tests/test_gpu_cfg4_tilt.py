"""cfg4 as BASELINE.json states it — 32 000 000 fluid particles, box 2400.6 x 150 m, gravity from the scripted tilt
trace (the MPU6050 stand-in: 15 deg, 8 s, re-sampled every 0.1 s of simulated time, pi_sph_fluid.c:455-461) — on ONE
MI355X, twice: as a single context and as FOUR slab contexts (the decomposition the 8-GPU run uses, halo exchange through
the host).  Gate G7 (N-GPU == 1-GPU): rho within 1e-5 while the two runs are comparable (300 steps); the un-compressed
lattice then falls onto the floor (75 m of water: |v| reaches 60 m/s at the bounce), the flow turns chaotic and only
aggregates and conservation are compared at step 2000 — and there ONE staged evaluation of the CPU oracle on slab 1's range
of the developed state (its owned columns + two columns of halo, ~8 000 000 particles) under the gravity vector of that step: the
slab's live rho (G1), p (G2) and the accelerations of its fused force pass (G3) within 1e-5, as tests/test_gpu_cfg3.py does for
cfg3 (reference: get_gravity pi_sph_fluid.c:431-464, calculate_density / _particle_pressure / _accelerations :263-373)."""
import os

import numpy as np
import pytest

from conftest import B_EOS, fused_step_vs_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5
THREADS = min(16, os.cpu_count() or 1)


def test_cfg4_tilt_four_slabs_equal_single_context(sph, orc, oracle):
    spec = sph.BLOCK_SCENES["cfg4"]
    box, x0, y0, nx, ny = spec
    prm = sph.default_params(box)
    walls = sph.scene_walls(prm)
    f = sph.block_range(prm, x0, y0, nx, ny, 0, nx)
    n = len(f)
    assert n == 32000000
    dt = float(np.float32(prm.dt))

    def trace():
        g = sph.GravitySource(sph.GRAVITY_TILT, 9.81)
        k = [0]

        def nxt(_):
            k[0] += 1
            return g.sample(k[0] * dt)
        return g.sample(0.0), nxt

    g0, grav1 = trace()
    _, grav2 = trace()
    parts = sph.slab.partition_block(prm, spec, 4)
    assert parts[0][0] == 0 and parts[-1][1] == sph.slab.grid_columns(prm)
    slabs = []
    for r, (c0, c1) in enumerate(parts):
        loc, ids = sph.slab.local_block_subset(sph, prm, spec, c0, c1)      # each slab generates only its own columns
        assert abs(len(loc) - n / 4) < 0.02 * n
        slabs.append(sph.slab.GpuSlab(sph, prm, None, walls, c0, c1, r > 0, r < 3, g0[0], g0[1], local=(loc, ids)))
    runner = sph.slab.SlabRunner(slabs, sph.slab.LocalTransport(slabs))
    with sph.Context(prm, f, walls, g0[0], g0[1]) as ctx:
        del f
        done = 0
        for k in (300, 2000):
            for _ in range(k - done):
                ctx.step(1, *grav1(0))
            ctx.sync()
            runner.step(k - done, gravity=grav2)
            for s in slabs:
                s.sync()
            done = k
            ref = ctx.read_particles()
            out, du, dv, seen = runner.gather_local(n, sph.PARTICLE)
            assert np.all(seen == 1)                                           # every particle owned exactly once
            assert ctx.out_of_domain() == 0
            if k == 300:
                assert np.max(np.abs(out["rho"] - ref["rho"]) / ref["rho"]) <= 1e-5
                assert max(np.abs(out["x"] - ref["x"]).max(), np.abs(out["y"] - ref["y"]).max()) <= 1e-5
                assert np.abs(ref["u"]).max() > 0.05                           # the tilted gravity is pushing sideways
            else:
                for fld in ("x", "y", "rho"):
                    a, b = float(out[fld].astype(np.float64).mean()), float(ref[fld].astype(np.float64).mean())
                    assert abs(a - b) <= 1e-4 * abs(b), (fld, a, b)
                assert slabs[0].rebuilds() == slabs[3].rebuilds() > 100        # all slabs rebuilt in the same steps
            del ref, seen
            if k == 300:
                del out, du, dv
        # ONE more step of the single context (the fused loop bench.py times at this size), under the next sample of the trace:
        # the state around it, for the oracle's step on slab 1's range below
        one_before, one_a_before = ctx.read_particles(), ctx.read_accel()
        g_next = grav1(0)
        ctx.step(1, *g_next)
        one_after, one_a_after = ctx.read_particles(), ctx.read_accel()
    # ---- the oracle on slab 1's range of the developed state (step 2000): owned columns + 2 columns of halo on each side,
    # under the gravity vector the last step was given ----
    replay = sph.GravitySource(sph.GRAVITY_TILT, 9.81)      # (the source holds its value for 0.1 s since ITS last reading, like the
    for j in range(2001):                                   # reference's polling thread :455-461: replay the readings of the run)
        gx, gy = replay.sample(j * dt)
    assert abs(gx) > 0.5                                                          # ~5.6 degrees of tilt by now
    half_dt = 0.5 * dt
    c0, c1 = parts[1]
    gc = sph.slab.global_columns(prm, out["x"])
    sel = np.nonzero((gc >= c0 - 2) & (gc < c1 + 2))[0]
    own = (gc[sel] >= c0) & (gc[sel] < c1)
    assert 6000000 < own.sum() < 10000000
    assert np.hypot(out["u"][sel], out["v"][sel]).max() > 5.0                     # the lattice has fallen: a developed state
    p = oracle.params(tuple(box))
    ob = walls.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    of = out[sel].view(orc.PARTICLE).copy()
    oracle.eval(p, of, ob, gx, gy, flags=1, threads=THREADS)                                   # rho from x(t)
    assert np.max(np.abs(out["rho"][sel][own] - of["rho"][own]) / of["rho"][own]) <= TOL      # G1 (through the slab's lists)
    of["rho"] = out["rho"][sel]
    oracle.eval(p, of, ob, gx, gy, flags=2, threads=THREADS)                                   # p from the GPU's rho
    assert np.max(np.abs(out["p"][sel][own] - of["p"][own]) / (of["p"][own] + B_EOS)) <= TOL  # G2
    # the force pass saw the half-kicked velocity (:328 reads u, v after :616): v_half = v - 0.5 DT a (:638)
    of["p"] = out["p"][sel]
    of["u"] = (out["u"][sel].astype(np.float64) - half_dt * du[sel].astype(np.float64)).astype(np.float32)
    of["v"] = (out["v"][sel].astype(np.float64) - half_dt * dv[sel].astype(np.float64)).astype(np.float32)
    odu, odv, sa = oracle.eval(p, of, ob, gx, gy, flags=4, threads=THREADS, want_sum_abs=True)
    err = np.hypot(du[sel] - odu, dv[sel] - odv) / (sa + 9.81)
    assert np.max(err[own]) <= TOL                                                            # G3 (fused force + kick pass)
    for s in slabs:
        s.close()
    del slabs, runner, out, du, dv, of, odu, odv, sa, err
    # ---- the fused step 2000 -> 2001 of the single context against ONE oracle step, on the same range (positions and half-kicked
    # velocities of every particle in it; rho, p, a and the full-step velocity where the neighbourhood is complete: the owned
    # columns): pi_sph_fluid.c:612-641 at the HBM-resident size ----
    gc = sph.slab.global_columns(prm, one_before["x"])
    sel = np.nonzero((gc >= c0 - 2) & (gc < c1 + 2))[0]
    own = (gc[sel] >= c0) & (gc[sel] < c1)
    worst = fused_step_vs_oracle(orc, oracle, p, ob, one_before[sel], (one_a_before[0][sel], one_a_before[1][sel]), one_after[sel],
                                 (one_a_after[0][sel], one_a_after[1][sel]), g_next, dt, own=own, threads=THREADS, tag="cfg4 slab 1 range")
    print("cfg4 @2000, one fused step / oracle on %d particles (%d with their whole neighbourhood), worst error over tolerance: %s"
          % (len(sel), int(own.sum()), {k: round(v, 3) for k, v in worst.items()}))


def test_cfg4_eight_slabs_equal_single_context(sph):
    """the partition the 8-GPU run uses: EIGHT slab contexts of 4 000 000 particles on one device against the single context,
    100 steps under the tilt trace: rho and x within 1e-5, every particle owned once, the slabs rebuild in the same steps."""
    spec = sph.BLOCK_SCENES["cfg4"]
    box, x0, y0, nx, ny = spec
    prm = sph.default_params(box)
    walls = sph.scene_walls(prm)
    dt = float(np.float32(prm.dt))
    g = sph.GravitySource(sph.GRAVITY_TILT, 9.81)
    gs = [g.sample(k * dt) for k in range(101)]
    parts = sph.slab.partition_block(prm, spec, 8)
    slabs = []
    for r, (c0, c1) in enumerate(parts):
        loc, ids = sph.slab.local_block_subset(sph, prm, spec, c0, c1)
        assert abs(len(loc) - 4000000) < 0.03 * 4000000
        slabs.append(sph.slab.GpuSlab(sph, prm, None, walls, c0, c1, r > 0, r < 7, gs[0][0], gs[0][1], local=(loc, ids)))
    runner = sph.slab.SlabRunner(slabs, sph.slab.LocalTransport(slabs))
    it = iter(gs[1:])
    runner.step(100, gravity=lambda _k: next(it))
    for s in slabs:
        s.sync()
    out, du, dv, seen = runner.gather_local(nx * ny, sph.PARTICLE)
    assert np.all(seen == 1)
    reb = [s.rebuilds() for s in slabs]
    assert len(set(reb)) == 1
    for s in slabs:
        s.close()
    del slabs, runner, du, dv, seen
    f = sph.block_range(prm, x0, y0, nx, ny, 0, nx)
    with sph.Context(prm, f, walls, gs[0][0], gs[0][1]) as ctx:
        del f
        for k in range(100):
            ctx.step(1, *gs[k + 1])
        ctx.sync()
        ref = ctx.read_particles()
    assert np.max(np.abs(out["rho"] - ref["rho"]) / ref["rho"]) <= 1e-5
    assert max(np.abs(out["x"] - ref["x"]).max(), np.abs(out["y"] - ref["y"]).max()) <= 1e-5
