"""cfg3 as BASELINE.json states it — 8 000 000 fluid particles, dam break in a 2400 x 60 m box, 4 x-slabs — on ONE MI355X:
four slab contexts (the decomposition the 4-GPU run uses; halo exchange through the host) against a single context.
Gate G7 (N-GPU == 1-GPU): rho and x within 1e-5 while the two runs are comparable (300 steps); at step 2000 every
particle is owned exactly once, all slabs have rebuilt in the same steps and the aggregates agree; and ONE staged
evaluation of the CPU oracle on slab 1's range of the developed state (its owned columns + two columns of halo): the
slab's live rho (G1), p (G2) and the accelerations of its fused force pass (G3) within 1e-5."""
import os

import numpy as np
import pytest

from conftest import B_EOS, GX, GY

pytestmark = pytest.mark.gpu
TOL = 1e-5
G = 9.81
THREADS = min(16, os.cpu_count() or 1)


def test_cfg3_four_slabs_equal_single_context_and_oracle(sph, orc, oracle):
    spec = sph.BLOCK_SCENES["cfg3"]
    box, x0, y0, nx, ny = spec
    prm = sph.default_params(box)
    walls = sph.scene_walls(prm)
    f = sph.block_range(prm, x0, y0, nx, ny, 0, nx)
    n = len(f)
    assert n == 8000000 and len(walls) == 65600
    parts = sph.slab.partition_block(prm, spec, 4)
    assert parts[0][0] == 0 and parts[-1][1] == sph.slab.grid_columns(prm)
    slabs = []
    for r, (c0, c1) in enumerate(parts):
        loc, ids = sph.slab.local_block_subset(sph, prm, spec, c0, c1)      # each slab generates only its own columns
        slabs.append(sph.slab.GpuSlab(sph, prm, None, walls, c0, c1, r > 0, r < 3, GX, GY, local=(loc, ids)))
    runner = sph.slab.SlabRunner(slabs, sph.slab.LocalTransport(slabs))
    half_dt = 0.5 * float(np.float32(prm.dt))
    with sph.Context(prm, f, walls, GX, GY) as ctx:
        del f
        done = 0
        for k in (300, 2000):
            ctx.step(k - done, GX, GY)
            ctx.sync()
            runner.step(k - done, GX, GY)
            for s in slabs:
                s.sync()
            done = k
            ref = ctx.read_particles()
            out, du, dv, seen = runner.gather_local(n, sph.PARTICLE)
            assert np.all(seen == 1)                                           # every particle owned exactly once
            assert ctx.out_of_domain() == 0
            if k == 300:
                assert np.max(np.abs(out["rho"] - ref["rho"]) / ref["rho"]) <= TOL
                assert max(np.abs(out["x"] - ref["x"]).max(), np.abs(out["y"] - ref["y"]).max()) <= TOL
            else:
                for fld in ("x", "y", "rho"):
                    a, b = float(out[fld].astype(np.float64).mean()), float(ref[fld].astype(np.float64).mean())
                    assert abs(a - b) <= 1e-4 * abs(b), (fld, a, b)
                reb = [s.rebuilds() for s in slabs]
                assert reb[0] == reb[1] == reb[2] == reb[3] > 50               # all slabs rebuilt in the same steps
                assert all(s.diagnostics()[:6] == (0, 0, 0, 0, 0, 0) for s in slabs)      # no tile on the direct path
            del ref
    # ---- the oracle on slab 1's range of the developed state: owned columns + 2 columns of halo on each side ----
    c0, c1 = parts[1]
    gc = sph.slab.global_columns(prm, out["x"])
    sel = np.nonzero((gc >= c0 - 2) & (gc < c1 + 2))[0]
    own = (gc[sel] >= c0) & (gc[sel] < c1)
    assert 1500000 < own.sum() < 2500000
    p = oracle.params(tuple(box))
    ob = walls.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    of = out[sel].view(orc.PARTICLE).copy()
    oracle.eval(p, of, ob, GX, GY, flags=1, threads=THREADS)                                   # rho from x(t)
    assert np.max(np.abs(out["rho"][sel][own] - of["rho"][own]) / of["rho"][own]) <= TOL      # G1 (through the slab's lists)
    of["rho"] = out["rho"][sel]
    oracle.eval(p, of, ob, GX, GY, flags=2, threads=THREADS)                                   # p from the GPU's rho
    assert np.max(np.abs(out["p"][sel][own] - of["p"][own]) / (of["p"][own] + B_EOS)) <= TOL  # G2
    # the force pass saw the half-kicked velocity (:328 reads u, v after :616): v_half = v - 0.5 DT a (:638)
    of["p"] = out["p"][sel]
    of["u"] = (out["u"][sel].astype(np.float64) - half_dt * du[sel].astype(np.float64)).astype(np.float32)
    of["v"] = (out["v"][sel].astype(np.float64) - half_dt * dv[sel].astype(np.float64)).astype(np.float32)
    odu, odv, sa = oracle.eval(p, of, ob, GX, GY, flags=4, threads=THREADS, want_sum_abs=True)
    err = np.hypot(du[sel] - odu, dv[sel] - odv) / (sa + G)
    assert np.max(err[own]) <= TOL                                                            # G3 (fused force + kick pass)
    for s in slabs:
        s.close()
