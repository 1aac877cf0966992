"""Parity gates of SURVEY.md §8c on a real MI355X, all through the C ABI (include/sph.h).

  G0 psi            |dpsi|/psi <= 1e-5
  G1 density        from identical f32 (x,y):                max |drho|/rho <= 1e-5
  G2 pressure       from identical f32 rho (uploaded):       |dp| <= 1e-5 (p + B)
  G3 acceleration   from identical f32 (x,y,u,v,rho,p):      |da| <= 1e-5 (sum_j |m_j temp_ij gradW_ij| + |g|)
  G4 end-to-end     rho -> p -> a all recomputed: loose by construction (B = 2.3e7 turns 1 ulp of rho into
                    ~10 Pa).  Calibrated per fixture against the reference's OWN self-consistency, i.e. its
                    as-shipped -Ofast build vs its -O2 build on the same state (tests/golden/manifest.json
                    "fast_vs_strict"): rms|da|/rms|a| <= max(2e-3, 2 x self), max|da| <= max(1 m/s^2, 4 x self)
  G5 trajectory     default scene: max|dx| <= 1e-5 m at step 100, <= 1e-3 m at step 1000
  G7 invariants     pair antisymmetry, rest state, determinism of read-back order

Expected values come from the golden fixtures (real reference) and, for inputs beyond the
reference's 65 534-particle limit, from the CPU oracle that is itself pinned bit-exactly to them.
"""
import json
import os

import numpy as np
import pytest

from conftest import B_EOS, GOLDEN, GX, GY, boundary_particles, load_golden, oracle_block_300, particles

MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


def self_consistency(name, suffix):
    """the reference's -Ofast build vs its -O2 build on this fixture (oracle/gen_golden.py)."""
    fv = MANIFEST["fixtures"][name]["fast_vs_strict"]
    return fv["k" + suffix[1:]] if suffix else fv

pytestmark = pytest.mark.gpu

TOL = 1e-5          # the tolerance BASELINE.json's north_star states
G = 9.81


def m_fluid(prm):
    return np.float32(prm.rho0) * np.float32(prm.vol)


def make_ctx(sph, orc, box, state, boundary_xy, variant):
    prm = sph.default_params(box)
    f = particles(orc, state, m_fluid(prm))
    b = boundary_particles(orc, boundary_xy)
    ctx = sph.Context(prm, f, b, GX, GY)
    ctx.set_variant(variant)
    return prm, f, ctx


def sum_abs_terms(oracle, box, f_with_rho_p, b_with_psi):
    p = oracle.params(box)
    f = f_with_rho_p.copy()
    du, dv, sa = oracle.eval(p, f, b_with_psi, GX, GY, flags=4, want_sum_abs=True)
    return du, dv, sa


FIXTURES = [("drop.npz", "_0"), ("drop.npz", "_100"), ("drop.npz", "_1000"), ("drop.npz", "_2000"),
            ("drop.npz", "_4000"), ("block.npz", ""), ("gas.npz", "")]


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("name,suffix", FIXTURES)
def test_staged_gates_on_golden(sph, orc, oracle, name, suffix, variant):
    g = load_golden(name)
    box = tuple(g["box"]) if "box" in g else (0.0, 4.0, 0.0, 2.0)
    state = g["state" + suffix]
    prm, f, ctx = make_ctx(sph, orc, box, state, g["boundary_xy"], variant)
    with ctx:
        # G0
        gb = ctx.read_boundary()
        assert np.array_equal(gb["x"], g["boundary_xy"][:, 0]) and np.array_equal(gb["y"], g["boundary_xy"][:, 1])
        assert np.max(np.abs(gb["m"] - g["psi"]) / g["psi"]) <= TOL
        # G1 (sph_create evaluates rho,p,a from the positions; also through the stage entry point)
        got = ctx.read_particles()
        assert np.array_equal(got["x"], state[:, 0]) and np.array_equal(got["v"], state[:, 3])   # original order
        rho_ref, p_ref = g["rho" + suffix], g["p" + suffix]
        assert np.max(np.abs(got["rho"] - rho_ref) / rho_ref) <= TOL
        # G4 end to end (loose)
        du, dv = ctx.read_accel()
        edu, edv = g["eval_du" + suffix], g["eval_dv" + suffix]
        da = np.hypot(du - edu, dv - edv)
        rms_a = np.sqrt(np.mean(edu.astype(np.float64) ** 2 + edv.astype(np.float64) ** 2))
        ref_self = self_consistency(name, suffix)
        assert np.sqrt(np.mean(da.astype(np.float64) ** 2)) / rms_a <= max(2e-3, 2 * ref_self["a_rms_rel"])
        assert da.max() <= max(1.0, 4 * ref_self["a_abs"])
        # G2: pressure from the reference's rho
        fin = particles(orc, state, m_fluid(prm), rho=rho_ref, p=np.zeros_like(p_ref))
        ctx.upload_state(fin)
        ctx.eval_pressure()
        got = ctx.read_particles()
        assert np.array_equal(got["rho"], rho_ref)
        assert np.max(np.abs(got["p"] - p_ref) / (p_ref + B_EOS)) <= TOL
        flip = (got["p"] > 0) != (p_ref > 0)
        if flip.any():   # clamp flags may differ only where |B((rho/rho0)^7-1)| < 32 Pa
            assert np.all(np.maximum(got["p"], p_ref)[flip] < 32.0)
        # G1 again via sph_eval_density on the uploaded state
        ctx.eval_density()
        got = ctx.read_particles()
        assert np.max(np.abs(got["rho"] - rho_ref) / rho_ref) <= TOL
        # G3: acceleration from the reference's x,y,u,v,rho,p
        fin = particles(orc, state, m_fluid(prm), rho=rho_ref, p=p_ref)
        ctx.upload_state(fin)
        ctx.eval_accel(GX, GY)
        du, dv = ctx.read_accel()
        bpsi = boundary_particles(orc, g["boundary_xy"], g["psi"])
        odu, odv, sa = sum_abs_terms(oracle, box, fin, bpsi)
        assert np.array_equal(odu.view(np.uint32), edu.view(np.uint32))    # the oracle IS the reference here
        da = np.hypot(du - edu, dv - edv)
        assert np.max(da / (sa + G)) <= TOL
        # informational figure of SURVEY.md: share of particles within 1e-5 |a|
        amag = np.hypot(edu, edv)
        share = np.mean(da <= 1e-5 * np.maximum(amag, 1e-30))
        assert share > 0.5
        assert ctx.out_of_domain() == 0


@pytest.mark.parametrize("variant", [0, 1])
def test_g5_short_trajectory(sph, orc, variant):
    g = load_golden("drop.npz")
    prm, f, ctx = make_ctx(sph, orc, (0.0, 4.0, 0.0, 2.0), g["state_0"], g["boundary_xy"], variant)
    with ctx:
        done = 0
        for k, tol in [(1, 1e-6), (10, 1e-6), (100, 1e-5), (1000, 1e-3)]:
            ctx.step(k - done, GX, GY)
            done = k
            ctx.sync()
            got = ctx.read_particles()
            st = g["state_%d" % k]
            dx = max(np.abs(got["x"] - st[:, 0]).max(), np.abs(got["y"] - st[:, 1]).max())
            assert dx <= tol, (k, dx)
        # beyond ~1500 steps trajectories decorrelate: aggregates only
        ctx.step(3000, GX, GY)
        ctx.sync()
        got = ctx.read_particles()
        st = g["state_4000"]
        assert abs(got["y"].mean() - st[:, 1].mean()) <= 0.05 * st[:, 1].mean()
        assert got["x"].min() > 0 and got["x"].max() < 4 and got["y"].min() > 0 and got["y"].max() < 2
        assert 0.095 <= got["y"].min() <= 0.110          # equilibrium wall stand-off
        max_rho, max_speed = ctx.stats()
        assert abs(max_rho - got["rho"].max()) <= 1e-3 * max_rho
        assert abs(max_speed - np.hypot(got["u"], got["v"]).max()) <= 1e-4 * max(max_speed, 1.0)
        assert abs(max_rho - g["rho_4000"].max()) <= 0.02 * g["rho_4000"].max()


@pytest.mark.parametrize("variant", [0, 1])
def test_block_trajectory_vs_oracle(sph, orc, oracle, variant):
    """300 steps of the 14 400-particle dam break from the lattice, against the oracle."""
    g = load_golden("block.npz")
    box = tuple(g["box"])
    xy = g["fluid_xy0"]
    state = np.concatenate([xy, np.zeros_like(xy)], 1)
    prm, f, ctx = make_ctx(sph, orc, box, state, g["boundary_xy"], variant)
    b = boundary_particles(orc, g["boundary_xy"], g["psi"])
    of = oracle_block_300(oracle, orc, f, b, box)
    with ctx:
        ctx.step(300, GX, GY)
        ctx.sync()
        got = ctx.read_particles()
        assert max(np.abs(got["x"] - of["x"]).max(), np.abs(got["y"] - of["y"]).max()) <= 1e-4
        assert np.max(np.abs(got["rho"] - of["rho"]) / of["rho"]) <= 1e-3


def test_developed_state_large_vs_oracle(sph, orc, oracle):
    """a 120 000-particle developed flow (beyond the reference's ushort limit): tile the block fixture 8x
    along x, evaluate once on GPU and oracle.  G1 and staged G3 at this size."""
    g = load_golden("block.npz")
    reps = 8
    w = 40.0
    st = np.concatenate([g["state"] + np.array([k * w, 0, 0, 0], np.float32) for k in range(reps)])
    box = (0.0, w * reps, 0.0, 8.0)
    prm, _, b = sph.scene_block(box, 0.3, 0.3, 1, 1)
    f = particles(orc, st, m_fluid(prm))
    p = oracle.params(box)
    ob = b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    of = f.copy()
    odu, odv, sa = oracle.eval(p, of, ob, GX, GY, want_sum_abs=True)
    with sph.Context(prm, f, b, GX, GY) as ctx:
        got = ctx.read_particles()
        assert np.max(np.abs(got["rho"] - of["rho"]) / of["rho"]) <= TOL
        ctx.upload_state(of)
        ctx.eval_accel(GX, GY)
        du, dv = ctx.read_accel()
        assert np.max(np.hypot(du - odu, dv - odv) / (sa + G)) <= TOL


@pytest.mark.parametrize("variant", [0, 1])
def test_g7_invariants_at_scale(sph, variant):
    """size-independent properties on a 500 000-particle block far from every wall (no oracle needed)."""
    prm, f, b = sph.scene_block((0.0, 300.0, 0.0, 60.0), 5.0, 5.0, 1000, 500)
    rng = np.random.default_rng(7)
    f["u"] = rng.normal(0, 0.5, len(f)).astype(np.float32)      # switch the viscosity branch on for ~half the pairs
    f["v"] = rng.normal(0, 0.5, len(f)).astype(np.float32)
    with sph.Context(prm, f, b, GX, GY) as ctx:
        ctx.set_variant(variant)
        ctx.step(20, GX, GY)
        ctx.sync()
        got = ctx.read_particles()
        # (a) order: read-back index i is the particle given as fluid[i]; it moved < 20 dt |v|max
        assert np.abs(got["x"] - f["x"]).max() < 0.05 and np.abs(got["y"] - f["y"]).max() < 0.05
        assert got["p"].max() > 0
        # (b) pair antisymmetry (momentum conservation): with g = 0 and no wall in reach,
        #     sum_i m a_i = 0 up to rounding: |sum a| <= 1e-5 sum_i sum_j |terms| ~ 1e-4 sum_i |a_i|
        ctx.upload_state(got)
        ctx.eval_accel(0.0, 0.0)
        du, dv = ctx.read_accel()
        scale = float(np.sum(np.hypot(du, dv).astype(np.float64)))
        assert scale > 0
        assert abs(float(np.sum(du.astype(np.float64)))) <= 1e-4 * scale
        assert abs(float(np.sum(dv.astype(np.float64)))) <= 1e-4 * scale
        # (c) idempotence: evaluating again on the same state gives the same answer to rounding
        ctx.eval_accel(0.0, 0.0)
        du2, dv2 = ctx.read_accel()
        assert np.max(np.abs(du2 - du)) <= 1e-4 * (np.abs(du).max() + 1)
        assert np.max(np.abs(dv2 - dv)) <= 1e-4 * (np.abs(dv).max() + 1)
        # (d) gravity enters additively (linearity in g)
        ctx.eval_accel(1.5, -2.5)
        du3, dv3 = ctx.read_accel()
        assert np.max(np.abs((du3 - du) - 1.5)) <= 1e-5 * (np.abs(du).max() + 1.5)
        assert np.max(np.abs((dv3 - dv) + 2.5)) <= 1e-5 * (np.abs(dv).max() + 2.5)


def test_g7_rest_lattice_far_from_walls(sph):
    """a resting lattice with g = 0: interior accelerations vanish by symmetry (free-surface rows excluded)."""
    prm, f, b = sph.scene_block((0.0, 30.0, 0.0, 30.0), 5.0, 5.0, 200, 200)
    with sph.Context(prm, f, b, 0.0, 0.0) as ctx:
        du, dv = ctx.read_accel()
        got = ctx.read_particles()
        inner = (f["x"] > 5.5) & (f["x"] < 19.4) & (f["y"] > 5.5) & (f["y"] < 19.4)
        a = np.hypot(du, dv)[inner]
        # pair terms are O(m k1 (W/W02)^4 |gradW|) ~ 1e2 m/s^2 each; their lattice sum cancels to rounding
        assert a.max() < 5e-3
        assert np.ptp(got["rho"][inner]) < 1e-4 * 973.0      # f32 lattice coordinates jitter by ~1 ulp


def test_variants_agree(sph):
    prm, f, b = sph.scene_block((0.0, 60.0, 0.0, 20.0), 0.3, 0.3, 400, 100)
    outs = []
    for v in (0, 1):
        with sph.Context(prm, f, b, GX, GY) as ctx:
            ctx.set_variant(v)
            ctx.step(50, GX, GY)
            ctx.sync()
            outs.append(ctx.read_particles())
    assert np.abs(outs[0]["x"] - outs[1]["x"]).max() <= 1e-5
    assert np.max(np.abs(outs[0]["rho"] - outs[1]["rho"]) / outs[1]["rho"]) <= 1e-4


def test_edge_cases(sph, orc):
    prm, f, b = sph.scene("cfg0")
    # empty fluid set
    with sph.Context(prm, f[:0], b, GX, GY) as ctx:
        ctx.step(3, GX, GY)
        ctx.sync()
        assert len(ctx.read_particles()) == 0
        assert np.all(ctx.read_boundary()["m"] > 0)
    # single particle, no boundary: rho = m W(0), a = g
    with sph.Context(prm, f[:1], b[:0], 0.25, -9.0) as ctx:
        got = ctx.read_particles()
        assert abs(got["rho"][0] - f["m"][0] * 58.597477) <= 1e-3
        du, dv = ctx.read_accel()
        assert du[0] == np.float32(0.25) and dv[0] == np.float32(-9.0)
    # ragged size (not a multiple of the workgroup) and coincident distinct particles: finite, no force between them
    ff = np.concatenate([f[:257], f[5:6]])
    with sph.Context(prm, ff, b, GX, GY) as ctx:
        du, dv = ctx.read_accel()
        assert np.all(np.isfinite(du)) and np.all(np.isfinite(dv))
    # a particle outside the box is clamped into an edge cell and reported, never UB
    fo = f.copy()
    fo["x"][0] = -0.5
    with pytest.raises(sph.SphError) as e:
        sph.Context(prm, fo, b, GX, GY)
    assert e.value.code == sph.SPH_E_OUT_OF_DOMAIN
    fo["x"][0] = np.nan
    with pytest.raises(sph.SphError) as e:
        sph.Context(prm, fo, b, GX, GY)
    assert e.value.code == sph.SPH_E_NAN


def test_time_varying_gravity(sph, orc, oracle):
    """gravity is sampled once per sph_step call (reference: every step, :632)."""
    prm, f, b = sph.scene("cfg0")
    p = oracle.params()
    of, ob = f.view(orc.PARTICLE).copy(), b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    grav = sph.GravitySource(sph.GRAVITY_TILT, 9.81, hold_s=0.0, period_s=0.05)
    g0 = grav.sample(0.0)
    du, dv = oracle.eval(p, of, ob, *g0)
    with sph.Context(prm, f, b, *g0) as ctx:
        t = 0.0
        for s in range(60):
            t += prm.dt
            gx, gy = grav.sample(t)
            ctx.step(1, gx, gy)
            oracle.steps(p, of, ob, gx, gy, du, dv, 1, threads=8)
        ctx.sync()
        got = ctx.read_particles()
        assert max(np.abs(got["x"] - of["x"]).max(), np.abs(got["y"] - of["y"]).max()) <= 1e-5
        gdu, gdv = ctx.read_accel()
        assert np.abs(gdu - du).max() <= 1e-3 and np.abs(gdv - dv).max() <= 1e-3


@pytest.mark.parametrize("k", [0, 1000, 4000])
def test_metaballs(sph, orc, k):
    """row f1: the 128x64 SSD1306 page-format bitmap of draw_metaballs (:380-411)."""
    g = load_golden("drop.npz")
    prm, f, ctx = make_ctx(sph, orc, (0.0, 4.0, 0.0, 2.0), g["state_%d" % k], g["boundary_xy"], 0)
    with ctx:
        got = np.unpackbits(ctx.render_metaballs())
        exp = np.unpackbits(g["metaballs_%d" % k])
        # pixels whose field value is within rounding of the threshold may flip; everything else is exact
        assert np.count_nonzero(got != exp) <= 2
        assert exp.sum() > 100


def _orc_params_from(oracle, prm):
    p = oracle.params((prm.x_min, prm.x_max, prm.y_min, prm.y_max))
    for k in ("r", "h", "rho0", "c", "g", "dt", "vol"):
        setattr(p, k, getattr(prm, k))
    p.alpha, p.eps, p.k1, p.k2 = float(prm.alpha), float(prm.eps), float(prm.k1), float(prm.k2)
    return p


@pytest.mark.parametrize("r,c_sound,alpha,k1", [(0.05, 300.0, 0.02, 0.05), (0.11, 600.0, 0.005, 0.2)])
def test_non_default_parameters(sph, orc, oracle, r, c_sound, alpha, k1):
    """every constant of pi_sph_fluid.c:11-20, :325-334 is a run-time parameter: other spacing, sound speed,
    viscosity and artificial-pressure strength against the oracle evaluated with the same parameters."""
    prm = sph.default_params((0.0, 10.0, 0.0, 5.0))
    prm.r = r
    prm.h = np.float32(r) * np.float32(1.3)
    prm.c = c_sound
    prm.dt = np.float32(prm.h) / np.float32(c_sound)
    prm.vol = np.float32(0.57) * np.float32(prm.h) * np.float32(prm.h)
    prm.alpha, prm.k1 = alpha, k1
    nx, ny = int(4.0 / r), int(2.0 / r)
    L = sph.host_lib()
    import ctypes as C
    f = np.zeros(nx * ny, sph.PARTICLE)
    assert L.sph_scene_block(C.byref(prm), 4 * r, 4 * r, nx, ny, f.ctypes.data_as(C.c_void_p), len(f)) == len(f)
    nb = L.sph_scene_walls(C.byref(prm), 0, None, 0)
    b = np.zeros(nb, sph.PARTICLE)
    L.sph_scene_walls(C.byref(prm), 0, b.ctypes.data_as(C.c_void_p), nb)
    rng = np.random.default_rng(11)
    f["u"] = rng.normal(0, 1.0, len(f)).astype(np.float32)
    f["v"] = rng.normal(0, 1.0, len(f)).astype(np.float32)
    p = _orc_params_from(oracle, prm)
    of, ob = f.view(orc.PARTICLE).copy(), b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    du, dv, sa = oracle.eval(p, of, ob, GX, GY, want_sum_abs=True)
    with sph.Context(prm, f, b, GX, GY) as ctx:
        assert ctx.grid_dims() == oracle.grid_dims(p)
        gb = ctx.read_boundary()
        assert np.max(np.abs(gb["m"] - ob["m"]) / ob["m"]) <= TOL
        got = ctx.read_particles()
        assert np.max(np.abs(got["rho"] - of["rho"]) / of["rho"]) <= TOL
        ctx.upload_state(of)
        ctx.eval_pressure()
        Bp = float(np.float32(c_sound) * np.float32(c_sound) * np.float32(1000.0) / np.float32(7))
        assert np.max(np.abs(ctx.read_particles()["p"] - of["p"]) / (of["p"] + Bp)) <= TOL
        ctx.upload_state(of)
        ctx.eval_accel(GX, GY)
        gdu, gdv = ctx.read_accel()
        assert np.max(np.hypot(gdu - du, gdv - dv) / (sa + G)) <= TOL
        ctx.upload_state(f)
        ctx.eval_density(); ctx.eval_pressure(); ctx.eval_accel(GX, GY)
        ctx.step(40, GX, GY)
        ctx.sync()
        got = ctx.read_particles()
    oracle.steps(p, of, ob, GX, GY, du, dv, 40, threads=8)
    assert max(np.abs(got["x"] - of["x"]).max(), np.abs(got["y"] - of["y"]).max()) <= 1e-5


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_clusters_vs_oracle(sph, orc, oracle, seed):
    """irregular inputs: clumps of very different density, empty regions, particles on cell edges and tile sizes
    that force every staging mode (one window, two windows, sparse) and the oversized-tile fallback."""
    rng = np.random.default_rng(seed)
    box = (0.0, 30.0, 0.0, 12.0)
    prm, _, b = sph.scene_block(box, 0.3, 0.3, 1, 1)
    pts = []
    for _ in range(12):                                    # gaussian clumps, some very dense
        cx, cy = rng.uniform(2, 28), rng.uniform(2, 10)
        n, s = int(rng.integers(200, 1500)), rng.uniform(0.4, 1.5)      # up to ~300 neighbours (oracle scratch: 512)
        pts.append(np.stack([rng.normal(cx, s, n), rng.normal(cy, s * 0.6, n)], 1))
    pts.append(np.stack([rng.uniform(0.2, 29.8, 3000), rng.uniform(0.2, 11.8, 3000)], 1))      # sparse spray
    cell = 0.195000008
    edge = np.stack([np.round(rng.uniform(1, 29, 400) / cell) * cell, rng.uniform(1, 11, 400)], 1)   # on cell edges
    pts.append(edge)
    xy = np.concatenate(pts).astype(np.float32)
    xy = xy[(xy[:, 0] > 0.1) & (xy[:, 0] < 29.9) & (xy[:, 1] > 0.1) & (xy[:, 1] < 11.9)]
    f = particles(orc, np.concatenate([xy, rng.normal(0, 2, xy.shape).astype(np.float32)], 1), m_fluid(prm))
    p = oracle.params(box)
    ob = b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    of = f.copy()
    du, dv, sa = oracle.eval(p, of, ob, GX, GY, want_sum_abs=True)
    for variant in (0, 1):
        with sph.Context(prm, f, b, GX, GY) as ctx:
            ctx.set_variant(variant)
            ctx.upload_state(f)
            ctx.eval_density()
            got = ctx.read_particles()
            assert np.max(np.abs(got["rho"] - of["rho"]) / of["rho"]) <= TOL, variant
            ctx.upload_state(of)
            ctx.eval_accel(GX, GY)
            gdu, gdv = ctx.read_accel()
            assert np.max(np.hypot(gdu - du, gdv - dv) / (sa + G)) <= TOL, variant


@pytest.mark.parametrize("variant", [0, 1])
def test_wall_velocity_enters_the_viscosity_term(sph, orc, oracle, variant):
    """the reference's fluid-boundary viscosity reads the wall particle's stored u, v (pi_sph_fluid.c:357); its own
    scenes leave them 0.  With non-zero wall velocities (a sliding floor) the staged G3 gate must still hold, and the
    result must differ from the wall-at-rest one."""
    g = load_golden("drop.npz")
    box = (0.0, 4.0, 0.0, 2.0)
    state, rho_ref, p_ref = g["state_2000"], g["rho_2000"], g["p_2000"]     # fluid resting on the floor
    prm = sph.default_params(box)
    b = boundary_particles(orc, g["boundary_xy"])
    b["u"] = np.where(g["boundary_xy"][:, 1] == 0.0, 3.0, 0.0)              # the floor slides at 3 m/s
    b["v"] = 0.25
    fin = particles(orc, state, m_fluid(prm), rho=rho_ref, p=p_ref)
    with sph.Context(prm, fin, b, GX, GY) as ctx:
        ctx.set_variant(variant)
        gb = ctx.read_boundary()
        assert np.array_equal(gb["u"], b["u"]) and np.array_equal(gb["v"], b["v"])
        ctx.upload_state(fin)
        ctx.eval_accel(GX, GY)
        du, dv = ctx.read_accel()
    bpsi = b.copy()
    bpsi["m"] = g["psi"]
    odu, odv, sa = sum_abs_terms(oracle, box, fin, bpsi)
    assert np.max(np.hypot(du - odu, dv - odv) / (sa + G)) <= TOL
    assert np.max(np.hypot(odu - g["eval_du_2000"], odv - g["eval_dv_2000"])) > 1e-2    # the wall velocity matters


def test_set_boundary_velocity_against_oracle(sph, orc, oracle):
    """sph_set_boundary_velocity: every wall particle gets (u, v) without re-binning (the velocity a host infers from its
    accelerometer, README.md:175-176): the staged G3 gate against the oracle run with the same wall velocities, the
    read-back of the walls, and a step afterwards still using them."""
    g = load_golden("drop.npz")
    box = (0.0, 4.0, 0.0, 2.0)
    state, rho_ref, p_ref = g["state_2000"], g["rho_2000"], g["p_2000"]     # fluid resting on the floor
    prm = sph.default_params(box)
    b = boundary_particles(orc, g["boundary_xy"])
    fin = particles(orc, state, m_fluid(prm), rho=rho_ref, p=p_ref)
    with sph.Context(prm, fin, b, GX, GY) as ctx:
        ctx.set_boundary_velocity(2.5, -0.4)
        gb = ctx.read_boundary()
        assert np.all(gb["u"] == np.float32(2.5)) and np.all(gb["v"] == np.float32(-0.4))
        assert np.array_equal(gb["x"], b["x"]) and np.array_equal(gb["y"], b["y"])      # nobody moved
        ctx.upload_state(fin)
        ctx.eval_accel(GX, GY)
        du, dv = ctx.read_accel()
        ctx.step(3, GX, GY)                                                   # (the step's force pass reads the same array)
        ctx.sync()
        assert np.all(ctx.read_boundary()["u"] == np.float32(2.5))
    bpsi = b.copy()
    bpsi["m"] = g["psi"]
    bpsi["u"] = 2.5
    bpsi["v"] = -0.4
    odu, odv, sa = sum_abs_terms(oracle, box, fin, bpsi)
    assert np.max(np.hypot(du - odu, dv - odv) / (sa + G)) <= TOL
    assert np.max(np.hypot(odu - g["eval_du_2000"], odv - g["eval_dv_2000"])) > 1e-2    # the wall velocity matters


def test_restart_from_read_back_continues_the_run(sph, orc):
    """checkpoint / resume: sph_upload_state(sph_read_particles()) keeps du_dt, dv_dt aligned with their particles (the
    reference's arrays are index-aligned, :616), so stepping on continues the uninterrupted run; a fresh context needs
    sph_upload_accel as well.  Both against the uninterrupted run (summation order only: the re-bin reorders lists)."""
    g = load_golden("block.npz")
    box = tuple(g["box"])
    prm, f, ctx = make_ctx(sph, orc, box, g["state"], g["boundary_xy"], 0)
    b = boundary_particles(orc, g["boundary_xy"])
    with ctx:
        ctx.step(37, GX, GY)
        ctx.sync()
        snap = ctx.read_particles()
        sdu, sdv = ctx.read_accel()
        ctx.step(25, GX, GY)
        ctx.sync()
        ref = ctx.read_particles()
    tol_x = 2e-5
    with sph.Context(prm, f, b, GX, GY) as c1:          # a different context restored from the checkpoint
        c1.upload_state(snap)
        c1.upload_accel(sdu, sdv)
        du, dv = c1.read_accel()
        assert np.array_equal(du, sdu) and np.array_equal(dv, sdv)
        c1.step(25, GX, GY)
        c1.sync()
        got = c1.read_particles()
        assert max(np.abs(got["x"] - ref["x"]).max(), np.abs(got["y"] - ref["y"]).max()) <= tol_x
        # upload_state alone: the stored accelerations follow their particles through the re-bin, bit for bit, and
        # the next step kicks every particle with ITS du_dt (:616, :622)
        p0 = c1.read_particles()
        a0u, a0v = c1.read_accel()
        c1.upload_state(p0)
        a1u, a1v = c1.read_accel()
        assert np.array_equal(a0u, a1u) and np.array_equal(a0v, a1v)
        c1.step(1, GX, GY)
        c1.sync()
        p1 = c1.read_particles()
        hdt, dt = 0.5 * float(np.float32(prm.dt)), float(np.float32(prm.dt))
        uh = (p0["u"].astype(np.float64) + hdt * a0u).astype(np.float32)
        vh = (p0["v"].astype(np.float64) + hdt * a0v).astype(np.float32)
        xp = (p0["x"].astype(np.float64) + dt * uh.astype(np.float64)).astype(np.float32)
        yp = (p0["y"].astype(np.float64) + dt * vh.astype(np.float64)).astype(np.float32)
        assert max(np.abs(p1["x"] - xp).max(), np.abs(p1["y"] - yp).max()) <= 1e-5     # 1 ulp at x ~ 40 m is 4e-6
    # a scrambled du_dt would show as O(dt^2 |a|) = 1e-4 m after one step and far more after 25


def test_time_kernel_needs_a_step(sph):
    """sph_time_kernel(force) repeats the last step's kick: refused before the first step (it would kick twice)."""
    prm, f, b = sph.scene("cfg0")
    with sph.Context(prm, f, b, GX, GY) as ctx:
        with pytest.raises(sph.SphError) as e:
            ctx.time_kernel("force_kick", 2)
        assert e.value.code == sph.SPH_E_STATE
        assert ctx.time_kernel("density_eos", 2) > 0
        ctx.step(3, GX, GY)
        before = ctx.read_particles()
        assert ctx.time_kernel("force_kick", 3) > 0
        after = ctx.read_particles()
        for k in ("x", "y", "u", "v", "rho"):
            assert np.array_equal(before[k], after[k]), k      # idempotent on the live state
        ctx.upload_state(before)
        with pytest.raises(sph.SphError):
            ctx.time_kernel("force_kick", 2)


def test_request_rebuild_and_timed_list_build(sph, orc, oracle):
    """sph_request_rebuild forces a rebuild in the next step whatever the criterion says; sph_time_kernel('build_list')
    rebuilds the lists in place and leaves the request raised.  Neither changes the trajectory beyond summation order."""
    g = load_golden("block.npz")
    prm = sph.default_params(tuple(g["box"]))
    f = particles(orc, g["state"], np.float32(prm.rho0) * np.float32(prm.vol))
    b = boundary_particles(orc, g["boundary_xy"])
    with sph.Context(prm, f, b, GX, GY) as ctx, sph.Context(prm, f, b, GX, GY) as ref:
        r0 = ctx.rebuild_stats()[0]
        for _ in range(5):
            ctx.request_rebuild()
            ctx.step(1, GX, GY)
        ctx.sync()
        assert ctx.rebuild_stats()[0] - r0 == 5
        assert ctx.time_kernel("build_list", 3) > 0
        r1 = ctx.rebuild_stats()[0]
        ctx.step(1, GX, GY)                       # finds the request raised: a full rebuild
        ctx.sync()
        assert ctx.rebuild_stats()[0] - r1 >= 1 and ctx.rebuild_stats()[1] == 0
        ref.step(6, GX, GY)
        ref.sync()
        a, r = ctx.read_particles(), ref.read_particles()
        assert max(np.abs(a["x"] - r["x"]).max(), np.abs(a["y"] - r["y"]).max()) <= 2e-5
        assert np.max(np.abs(a["rho"] - r["rho"]) / r["rho"]) <= 2e-4
