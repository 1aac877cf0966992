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
