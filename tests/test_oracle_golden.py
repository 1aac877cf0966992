"""Gate G6: the CPU oracle (oracle/sph_oracle.c, -O2) is BIT-EXACT against the golden vectors, which
were produced by the real reference's own functions (oracle/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, GX, GY, bits_equal, boundary_particles, load_golden, particles

M_FLUID = np.float32(1000.0) * (np.float32(0.57) * np.float32(0.0975000039) * np.float32(0.0975000039))


def test_manifest_hashes():
    import hashlib
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    for name, meta in man["fixtures"].items():
        with open(os.path.join(GOLDEN, name), "rb") as fh:
            assert hashlib.sha256(fh.read()).hexdigest() == meta["sha256"], name


def test_constants_bit_patterns(oracle):
    g = load_golden("drop.npz")
    p = oracle.params()
    assert bits_equal(oracle.constants(p)[:15], g["constants"][:15])
    # the values SURVEY.md §8a (a1) quotes
    c = oracle.constants(p)
    assert c[1] == np.float32(0.0975000039) and c[7] == np.float32(2.43750008e-4)
    assert c[10] == np.float32(22857142.0) and abs(c[11] - 58.597477) < 1e-5 and abs(c[12] - 53.8241196) < 1e-5


def test_default_scene_and_psi(oracle):
    g = load_golden("drop.npz")
    p = oracle.params()
    f, b = oracle.scene_default(p)
    assert len(f) == 269 and len(b) == 162
    assert bits_equal(np.stack([f["x"], f["y"]], 1), g["fluid_xy0"])
    assert bits_equal(np.stack([b["x"], b["y"]], 1), g["boundary_xy"])
    assert bits_equal(f["m"], np.full(269, M_FLUID, np.float32))
    oracle.psi(p, b)
    assert bits_equal(b["m"], g["psi"])
    # known answers of SURVEY.md §8a (a14)
    assert abs(b["m"].astype(np.float64).sum() - 3561.153946) < 1e-5
    assert abs(b["m"][0] - 9.744979) < 1e-5


@pytest.mark.parametrize("k", [0, 1, 10, 100, 1000, 2000, 4000])
def test_drop_stage_outputs(oracle, orc, k):
    """rho, p, a evaluated by the oracle on the reference's state at step k."""
    g = load_golden("drop.npz")
    p = oracle.params()
    b = boundary_particles(orc, g["boundary_xy"], g["psi"])
    f = particles(orc, g["state_%d" % k], M_FLUID)
    du, dv = oracle.eval(p, f, b, GX, GY, threads=4)
    assert bits_equal(f["rho"], g["rho_%d" % k])
    assert bits_equal(f["p"], g["p_%d" % k])
    assert bits_equal(du, g["eval_du_%d" % k]) and bits_equal(dv, g["eval_dv_%d" % k])
    if k == 0:
        assert abs(f["rho"].astype(np.float64).sum() - 248186.3355) < 1e-3


def test_drop_trajectory(oracle, orc):
    """the integrator loop :612-641: state after k steps is bit-identical to the reference's."""
    g = load_golden("drop.npz")
    p = oracle.params()
    b = boundary_particles(orc, g["boundary_xy"], g["psi"])
    f = particles(orc, g["state_0"], M_FLUID)
    du, dv = oracle.eval(p, f, b, GX, GY, threads=4)
    k_now = 0
    for k in [1, 10, 100, 1000, 2000, 4000]:
        oracle.steps(p, f, b, GX, GY, du, dv, k - k_now, threads=4)
        k_now = k
        st = np.stack([f["x"], f["y"], f["u"], f["v"]], 1)
        assert bits_equal(st, g["state_%d" % k]), k
        assert bits_equal(du, g["du_%d" % k]), k


@pytest.mark.parametrize("name", ["block.npz", "gas.npz"])
def test_block_and_gas(oracle, orc, name):
    g = load_golden(name)
    p = oracle.params(g["box"])
    b = boundary_particles(orc, g["boundary_xy"])
    oracle.psi(p, b)
    assert bits_equal(b["m"], g["psi"])
    f = particles(orc, g["state"], M_FLUID)
    du, dv = oracle.eval(p, f, b, GX, GY)
    assert bits_equal(f["rho"], g["rho"]) and bits_equal(f["p"], g["p"])
    assert bits_equal(du, g["eval_du"]) and bits_equal(dv, g["eval_dv"])


def test_block_trajectory_from_lattice(oracle, orc):
    """3000 steps of the dam-break block from its lattice reproduce the fixture's developed state."""
    g = load_golden("block.npz")
    p = oracle.params(g["box"])
    b = boundary_particles(orc, g["boundary_xy"], g["psi"])
    xy = g["fluid_xy0"]
    f = particles(orc, np.concatenate([xy, np.zeros_like(xy)], 1), M_FLUID)
    du, dv = oracle.eval(p, f, b, GX, GY)
    oracle.steps(p, f, b, GX, GY, du, dv, int(g["nsteps"]))
    assert bits_equal(np.stack([f["x"], f["y"], f["u"], f["v"]], 1), g["state"])


@pytest.mark.parametrize("k", [0, 1000, 4000])
def test_metaballs(oracle, orc, k):
    g = load_golden("drop.npz")
    p = oracle.params()
    f = particles(orc, g["state_%d" % k], M_FLUID)
    assert np.array_equal(oracle.metaballs(p, f, threads=4), g["metaballs_%d" % k])


def test_thread_count_independence(oracle, orc):
    """results do not depend on the OpenMP team size."""
    g = load_golden("block.npz")
    p = oracle.params(g["box"])
    b = boundary_particles(orc, g["boundary_xy"], g["psi"])
    outs = []
    for t in (1, 3, 8):
        f = particles(orc, g["state"], M_FLUID)
        du, dv = oracle.eval(p, f, b, GX, GY, threads=t)
        outs.append((f["rho"].copy(), du, dv))
    for o in outs[1:]:
        assert all(bits_equal(a, c) for a, c in zip(o, outs[0]))
