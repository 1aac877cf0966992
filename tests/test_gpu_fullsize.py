"""Parity at the sizes bench.py times: the full cfg1 (262 144-particle drop, 2.2 M-cell grid that is > 95 % empty) and
cfg2 (2 000 000-particle dam break) scenes, at t = 0 and on a GPU-advanced developed state (cfg1: through the impact of
the drop; cfg2: 4000 steps, the window with splashes, direct tiles and frequent rebuilds).

The reference cannot run these (unsigned short indices, pi_sph_fluid.c:78-79); the expected values are ONE evaluation
of the CPU oracle per stage at full size (32-bit indices; bit-pinned to the compiled reference at <= 65 534 particles,
tests/test_oracle_vs_ref.py).  Gates as in test_gpu_parity.py, tolerance 1e-5:

  live   the state the step loop left behind, checked as it is: rho(t) from x(t) through neighbour lists that are up
         to a few steps old (G1), p from that rho (G2), a(t) from x, v_half, rho, p of the same step (G3)
  staged upload -> re-bin -> sph_eval_* at the developed positions: G1, G2, G3 with the oracle's rho / p as inputs
  exact  the list kernels (variant 0) against the exact 5x5-cell walk (variant 1) on the same state
  sum    sum_i a_i agrees with the oracle's to 1e-5 of sum_i sum_j |terms|
"""
import os

import numpy as np
import pytest

from conftest import B_EOS, GX, GY, fused_step_vs_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-5
G = 9.81
THREADS = min(16, os.cpu_count() or 1)      # the GPU box gives one GPU a 16-core share

# (scene, steps before the check).  cfg1: the drop's lowest point starts 9.285 m above the floor: impact at step ~5650.
CASES = [("cfg1", 0), ("cfg1", 7000), ("cfg2", 0), ("cfg2", 4000)]


def _scene(sph, name):
    return sph.dam_break(1) if name == "cfg2" else sph.scene(name)


@pytest.mark.parametrize("name,warm", CASES)
def test_fullsize_scene_vs_oracle(sph, orc, oracle, name, warm):
    prm, f, b = _scene(sph, name)
    box = (prm.x_min, prm.x_max, prm.y_min, prm.y_max)
    p = oracle.params(box)
    ob = b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    half_dt = 0.5 * float(np.float32(prm.dt))
    with sph.Context(prm, f, b, GX, GY) as ctx:
        # G0 at full size (16 386 / 33 600 wall particles)
        gb = ctx.read_boundary()
        assert np.max(np.abs(gb["m"] - ob["m"]) / ob["m"]) <= TOL
        if warm:
            ctx.step(warm, GX, GY)
            ctx.sync()
        rebuilds, direct = ctx.rebuild_stats()
        got = ctx.read_particles()
        du, dv = ctx.read_accel()
        assert ctx.out_of_domain() == 0
        assert np.all(np.isfinite(got["x"])) and np.all(np.isfinite(du))
        if warm:
            assert rebuilds > 1 and rebuilds < warm          # lists were reused, and rebuilt when needed
            assert np.hypot(got["u"], got["v"]).max() > 5.0  # the flow has developed

        # ---- live state: what the step loop computed, as it is ----
        of = got.view(orc.PARTICLE).copy()
        oracle.eval(p, of, ob, GX, GY, flags=1, threads=THREADS)                                   # rho from x(t)
        assert np.max(np.abs(got["rho"] - of["rho"]) / of["rho"]) <= TOL          # G1 (through the Verlet lists)
        of["rho"] = got["rho"]
        oracle.eval(p, of, ob, GX, GY, flags=2, threads=THREADS)                                   # p from the GPU's rho
        assert np.max(np.abs(got["p"] - of["p"]) / (of["p"] + B_EOS)) <= TOL      # G2
        flip = (got["p"] > 0) != (of["p"] > 0)
        assert np.all(np.maximum(got["p"], of["p"])[flip] < 32.0)
        # the force pass saw the half-kicked velocity (:328 reads u, v after :616): v_half = v - 0.5 DT a (:638)
        # (at t = 0 no kick has happened yet: the init sequence :607 evaluates a from u, v as given)
        of["p"] = got["p"]
        if warm:
            of["u"] = (got["u"].astype(np.float64) - half_dt * du.astype(np.float64)).astype(np.float32)
            of["v"] = (got["v"].astype(np.float64) - half_dt * dv.astype(np.float64)).astype(np.float32)
        odu, odv, sa = oracle.eval(p, of, ob, GX, GY, flags=4, threads=THREADS, want_sum_abs=True)
        assert np.max(np.hypot(du - odu, dv - odv) / (sa + G)) <= TOL             # G3 (fused force + kick pass)
        tot = float(np.sum(sa.astype(np.float64))) + G * len(f)
        assert abs(float(np.sum(du.astype(np.float64) - odu))) <= TOL * tot       # sum of a
        assert abs(float(np.sum(dv.astype(np.float64) - odv))) <= TOL * tot

        # ---- staged: re-bin at these positions, every stage from the oracle's inputs ----
        of = got.view(orc.PARTICLE).copy()
        oracle.eval(p, of, ob, GX, GY, flags=3, threads=THREADS)                                   # oracle rho, p
        fin = of.copy()
        fin["p"] = 0
        ctx.upload_state(fin)
        ctx.eval_density()
        g1 = ctx.read_particles()
        assert np.array_equal(g1["x"], got["x"]) and np.array_equal(g1["v"], got["v"])   # original order survives
        assert np.max(np.abs(g1["rho"] - of["rho"]) / of["rho"]) <= TOL           # G1 on fresh lists
        ctx.upload_state(fin)
        ctx.eval_pressure()
        g2 = ctx.read_particles()
        assert np.max(np.abs(g2["p"] - of["p"]) / (of["p"] + B_EOS)) <= TOL       # G2
        odu, odv, sa = oracle.eval(p, of, ob, GX, GY, flags=4, threads=THREADS, want_sum_abs=True)
        ctx.upload_state(of)
        ctx.eval_accel(GX, GY)
        sdu, sdv = ctx.read_accel()
        assert np.max(np.hypot(sdu - odu, sdv - odv) / (sa + G)) <= TOL           # G3
        _, direct2 = ctx.rebuild_stats()
        # Tiles on the direct (no list) path hold their kernels back.  The dam break and the falling drop must have none;
        # in the splash of the drop (|v| to 100 m/s, spray over hundreds of cell rows) a tile of 256 droplets may span more
        # rows than the list build's tables hold: that — and only that — is allowed, a fraction of a tile per rebuild
        why = ctx.direct_tile_reasons()
        if name == "cfg2" or warm == 0:
            assert direct2 == 0, (direct2, why)
        else:
            assert direct2 <= 0.5 * rebuilds and why[3] == 0 and why[4] == 0 and why[5] == 0, (direct2, rebuilds, why)

        # ---- exact walk (variant 1) on the same state ----
        ctx.set_variant(1)
        ctx.eval_density()
        e1 = ctx.read_particles()
        assert np.max(np.abs(e1["rho"] - g1["rho"]) / g1["rho"]) <= 2e-6
        ctx.upload_state(of)
        ctx.eval_accel(GX, GY)
        edu, edv = ctx.read_accel()
        assert np.max(np.hypot(edu - sdu, edv - sdv) / (sa + G)) <= 4e-6      # (two summation orders of ~30 f32 terms)
    print("%s @%d: rebuilds %d, direct tiles %d (live) / %d (after re-bin), max speed %.1f m/s"
          % (name, warm, rebuilds, direct, direct2, float(np.hypot(got["u"], got["v"]).max())))


@pytest.mark.parametrize("name,warm", [("cfg1", 7000), ("cfg2", 4000)])
def test_fullsize_one_fused_step_vs_oracle(sph, orc, oracle, name, warm):
    """What bench.py times is the fused loop: the force pass (k_force_list<KICK_DRIFT>) also makes the NEXT step's kick 1/2 +
    drift (pi_sph_fluid.c:615-624) and the second half kick (:637-640), and stores neither the acceleration nor the velocity
    between steps.  One step of that loop at full size, on the developed state, against ONE oracle step from the same state
    (round-4 verdict: its integration had met the oracle only through trajectories of <= 14 400 particles): positions,
    half-kicked and full-step velocities, rho / p / a of the new state, and that sph_read_accel's recomputed a is the a the
    kernel kicked with (conftest.fused_step_vs_oracle).  Then the same step from an UPLOADED copy of the state (sph_upload_state +
    sph_upload_accel: the first step after an upload drifts with the stand-alone k_kick_drift, the one after it with the
    look-ahead again): both against the oracle and against each other."""
    prm, f, b = _scene(sph, name)
    p = oracle.params((prm.x_min, prm.x_max, prm.y_min, prm.y_max))
    ob = b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    dt = float(np.float32(prm.dt))
    with sph.Context(prm, f, b, GX, GY) as ctx:
        ctx.step(warm, GX, GY)
        before, a_before = ctx.read_particles(), ctx.read_accel()
        assert np.hypot(before["u"], before["v"]).max() > 5.0
        ctx.step(1, GX, GY)                                   # consumes the look-ahead of step `warm`'s force pass
        after, a_after = ctx.read_particles(), ctx.read_accel()
        w1 = fused_step_vs_oracle(orc, oracle, p, ob, before, a_before, after, a_after, (GX, GY), dt, threads=THREADS, tag=name + " live")
        ctx.step(1, GX, GY)                                   # ... and once more: a step whose predecessor was read back in between
        after2, a_after2 = ctx.read_particles(), ctx.read_accel()
        w2 = fused_step_vs_oracle(orc, oracle, p, ob, after, a_after, after2, a_after2, (GX, GY), dt, threads=THREADS, tag=name + " live + 1")
    # the checkpoint path: a fresh context, the state uploaded (x, v, rho, p as read, du_dt, dv_dt), two steps
    with sph.Context(prm, f, b, GX, GY) as ctx:
        ctx.upload_state(before)
        ctx.upload_accel(*a_before)
        ctx.step(1, GX, GY)                                   # stand-alone kick 1/2 + drift (k_kick_drift), then density + fused force
        up, a_up = ctx.read_particles(), ctx.read_accel()
        w3 = fused_step_vs_oracle(orc, oracle, p, ob, before, a_before, up, a_up, (GX, GY), dt, threads=THREADS, tag=name + " uploaded")
        ctx.step(1, GX, GY)
        up2, a_up2 = ctx.read_particles(), ctx.read_accel()
        w4 = fused_step_vs_oracle(orc, oracle, p, ob, up, a_up, up2, a_up2, (GX, GY), dt, threads=THREADS, tag=name + " uploaded + 1")
    # the two routes to step warm + 1 agree: identical positions up to the rounding of one fma against a multiply + add
    for c in ("x", "y"):
        assert np.max(np.abs(up[c].astype(np.float64) - after[c].astype(np.float64)) / np.spacing(np.abs(after[c]))) <= 2.0
    assert np.max(np.abs(up["rho"] - after["rho"]) / after["rho"]) <= 4e-6
    print("%s @%d one fused step / oracle, worst error over tolerance: live %s | live + 1 %s | uploaded %s | uploaded + 1 %s"
          % (name, warm, *({k: round(v, 3) for k, v in w.items()} for w in (w1, w2, w3, w4))))
