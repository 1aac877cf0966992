"""Shared fixtures.  `-m "not gpu"` runs here (no GPU); `-m gpu` runs on a real MI355X."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GX, GY = 0.0, -9.81
B_EOS = 22857142.0   # C^2 RHO_0 / 7 (pi_sph_fluid.c:297)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sph():
    """The product package (ctypes mirror of include/sph.h + include/sph_host.h), built if needed."""
    mod = importlib.import_module("pi-sph-fluid_amd")
    if not (os.path.exists(mod.LIB_HIP) and os.path.exists(mod.LIB_HOST)):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    return _orc


@pytest.fixture(scope="session")
def oracle(orc):
    return orc.Oracle("strict")


@pytest.fixture(scope="session")
def reference(orc):
    """The real reference's hot path (oracle/_ref); skip when it was not built (no /root/reference)."""
    if not orc.Reference.available("strict"):
        subprocess.call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")])
    if not orc.Reference.available("strict"):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return orc.Reference("strict")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def particles(orc_mod, state, m, rho=None, p=None, rho0=1000.0):
    """struct particle array from an (n,4) x,y,u,v state."""
    f = np.zeros(len(state), orc_mod.PARTICLE)
    f["x"], f["y"], f["u"], f["v"] = state[:, 0], state[:, 1], state[:, 2], state[:, 3]
    f["m"] = m
    f["rho"] = rho0 if rho is None else rho
    if p is not None:
        f["p"] = p
    return f


def boundary_particles(orc_mod, xy, psi=None, rho0=1000.0):
    b = np.zeros(len(xy), orc_mod.PARTICLE)
    b["x"], b["y"] = xy[:, 0], xy[:, 1]
    b["rho"] = rho0
    if psi is not None:
        b["m"] = psi
    return b


def bits_equal(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


_block_cache = {}


def oracle_block_300(oracle, orc_mod, fluid, boundary_psi, box):
    """the oracle's state after 300 steps of the 14 400-particle dam break from the lattice (shared by several GPU tests:
    a few seconds of CPU each time otherwise)."""
    key = (len(fluid), tuple(box))
    if key not in _block_cache:
        p = oracle.params(box)
        of = fluid.copy()
        du, dv = oracle.eval(p, of, boundary_psi, GX, GY, threads=8)
        oracle.steps(p, of, boundary_psi, GX, GY, du, dv, 300, threads=8)
        _block_cache[key] = of
    return _block_cache[key].copy()


def fused_step_vs_oracle(orc_mod, oracle, p, ob, before, a_before, after, a_after, g_after, dt, own=None, threads=8, tol=1e-5,
                         tag=""):
    """ONE time step of the fused loop (pi_sph_fluid.c:612-641) against the oracle at whatever size the arrays have.

    before / after = sph_read_particles() around one sph_step(ctx, g_after, 1), a_before / a_after = sph_read_accel() at the same
    two points (original order; `own`: the particles whose whole neighbourhood is in the arrays — a slab's owned range inside a
    range with two columns of halo — default all).  What it pins, and against what:
      x, y          the kick 1/2 + drift (:615-624) that the GPU's force pass of the PREVIOUS step made as a look-ahead, against
                    oracle.steps(1) from the same (x, v, a): <= 2 ulp of the position + DT^2 tol (sum|terms| + g)
      v_half        v_after - DT/2 a_after against v_before + 0.5 DT a_before (:616, double product): ulps of v + DT tol (...)
      rho, p, a     G1 / G2 / G3 of the new state (oracle on the GPU's x_after; p from its rho; a from its x, v_half, rho, p)
      v             the full-step velocity (:638-639) against v_half + 0.5 DT a_oracle: ulps + DT/2 tol (sum|terms| + g)
      a recomputed  (v_after - v_before) / (DT/2) - a_after - a_before = 2 (a the kick used - a read back): the acceleration that
                    sph_read_accel re-evaluates IS the one the fused kernel kicked with (it does not store it), to rounding
    Returns the worst value of every check divided by its tolerance (all <= 1)."""
    P = orc_mod.PARTICLE
    G = float(np.hypot(*g_after))
    half = 0.5 * dt
    own = np.ones(len(before), bool) if own is None else own
    du1, dv1 = a_before
    du2, dv2 = a_after
    f64 = np.float64
    # sum_j |m_j temp_ij grad W_ij| of the state before: the scale of G3's tolerance for a_before (one oracle evaluation)
    s1 = before.view(P).copy()
    s1["u"] = (before["u"].astype(f64) - half * du1.astype(f64)).astype(np.float32)      # v_half of the step that made a_before
    s1["v"] = (before["v"].astype(f64) - half * dv1.astype(f64)).astype(np.float32)
    _, _, sa1 = oracle.eval(p, s1, ob, g_after[0], g_after[1], flags=4, threads=threads, want_sum_abs=True)
    del s1
    # the oracle's step from the same state
    of = before.view(P).copy()
    odu, odv = du1.copy(), dv1.copy()
    oracle.steps(p, of, ob, g_after[0], g_after[1], odu, odv, 1, threads=threads)
    worst = {}

    def gate(name, err, tolv, mask=None):
        r = err / tolv
        r = r if mask is None else r[mask]
        worst[name] = float(np.max(r))
        assert worst[name] <= 1.0, (tag, name, worst[name], int(np.argmax(r)))

    ulp = lambda a: np.spacing(np.abs(a).astype(np.float32)).astype(f64)
    scale1 = (sa1.astype(f64) + G)
    for c in ("x", "y"):      # :616 + :622 — depend on the inputs alone: every particle, halo or not
        gate(c, np.abs(after[c].astype(f64) - of[c].astype(f64)), 2 * ulp(of[c]) + dt * dt * tol * scale1)
    vh_o = {"u": (before["u"].astype(f64) + half * du1.astype(f64)).astype(np.float32),       # u += 0.5*DT*du_dt  :616
            "v": (before["v"].astype(f64) + half * dv1.astype(f64)).astype(np.float32)}
    vh_g = {"u": after["u"].astype(f64) - half * du2.astype(f64), "v": after["v"].astype(f64) - half * dv2.astype(f64)}
    for c in ("u", "v"):
        gate(c + "_half", np.abs(vh_g[c] - vh_o[c].astype(f64)), 3 * ulp(np.maximum(np.abs(after[c]), np.abs(vh_o[c]))) + dt * tol * scale1)
    # the new state: rho from the GPU's positions (G1), p from its rho (G2), a from its x, v_half, rho, p (G3)
    s2 = after.view(P).copy()
    oracle.eval(p, s2, ob, g_after[0], g_after[1], flags=1, threads=threads)
    gate("rho", np.abs(after["rho"] - s2["rho"]) / s2["rho"], tol, own)
    s2["rho"] = after["rho"]
    oracle.eval(p, s2, ob, g_after[0], g_after[1], flags=2, threads=threads)
    gate("p", np.abs(after["p"] - s2["p"]) / (s2["p"] + B_EOS), tol, own)
    s2["p"] = after["p"]
    s2["u"], s2["v"] = vh_g["u"].astype(np.float32), vh_g["v"].astype(np.float32)
    adu, adv, sa2 = oracle.eval(p, s2, ob, g_after[0], g_after[1], flags=4, threads=threads, want_sum_abs=True)
    scale2 = sa2.astype(f64) + G
    gate("a", np.hypot(du2 - adu, dv2 - adv) / scale2, tol, own)
    for c, a_o in (("u", adu), ("v", adv)):      # :638-639 from the oracle's a of the same inputs
        v_o = (vh_o[c].astype(f64) + half * a_o.astype(f64)).astype(np.float32)
        gate(c, np.abs(after[c].astype(f64) - v_o.astype(f64)), 3 * ulp(np.maximum(np.abs(after[c]), np.abs(v_o))) + half * tol * (scale1 + scale2), own)
    # the acceleration the kick used vs the one sph_read_accel recomputes (the fused kernel does not store it)
    for c, a1, a2 in (("u", du1, du2), ("v", dv1, dv2)):
        used_minus_read = 0.5 * ((after[c].astype(f64) - before[c].astype(f64)) / half - a2.astype(f64) - a1.astype(f64))
        gate("a_used_" + c, np.abs(used_minus_read), tol * scale1 + 4 * ulp(np.maximum(np.abs(after[c]), np.abs(before[c]))) / half)
    return worst
