"""The oracle against the LIVE compiled reference (oracle/_ref, built from /root/reference by
oracle/build_ref.sh) on inputs the fixtures do not hold.  Skipped where the reference is absent."""
import numpy as np
import pytest

from conftest import GX, GY, bits_equal

M_FLUID = np.float32(1000.0) * (np.float32(0.57) * np.float32(0.0975000039) * np.float32(0.0975000039))


def rand_scene(orc, seed, n, box):
    rng = np.random.default_rng(seed)
    f = np.zeros(n, orc.PARTICLE)
    # jittered lattice at the reference number density keeps neighbour counts <= 48 (the reference's scratch, :21)
    side = int(np.ceil(np.sqrt(n)))
    ii, jj = np.divmod(np.arange(n), side)
    f["x"] = (0.4 + 0.075 * ii + rng.uniform(-0.03, 0.03, n)).astype(np.float32)
    f["y"] = (0.4 + 0.075 * jj + rng.uniform(-0.03, 0.03, n)).astype(np.float32)
    f["u"] = rng.normal(0, 1.5, n).astype(np.float32)
    f["v"] = rng.normal(0, 1.5, n).astype(np.float32)
    f["m"] = M_FLUID
    f["rho"] = 1000.0
    xs = np.arange(0, box[1], 0.075, dtype=np.float32)
    ys = np.arange(0, box[3], 0.075, dtype=np.float32)
    b = np.zeros(2 * len(xs) + 2 * len(ys), orc.PARTICLE)
    b["x"] = np.concatenate([xs, xs, np.zeros_like(ys), np.full_like(ys, box[1])])
    b["y"] = np.concatenate([np.zeros_like(xs), np.full_like(xs, box[3]), ys, ys])
    b["rho"] = 1000.0
    return f, b


@pytest.mark.parametrize("seed,n,box", [(1, 900, (0.0, 5.0, 0.0, 5.0)), (2, 4096, (0.0, 7.0, 0.0, 7.0)), (3, 1, (0.0, 3.0, 0.0, 3.0))])
def test_eval_and_steps_bit_exact(oracle, reference, orc, seed, n, box):
    f, b = rand_scene(orc, seed, n, box)
    rf, rb = f.copy(), b.copy()
    p = oracle.params(box)
    rbox = reference.box(box)
    assert oracle.grid_dims(p) == reference.grid_dims(rbox)
    oracle.psi(p, b)
    reference.psi(rb, rbox)
    assert bits_equal(b["m"], rb["m"])
    mff, mfb = reference.max_neighbors(rf, rb, rbox)
    assert mff <= 48 and mfb <= 48
    assert oracle.max_neighbors(p, f, b) == (mff, mfb)
    du, dv = oracle.eval(p, f, b, 0.3, -9.0, threads=4)
    rdu, rdv = reference.eval(rf, rb, rbox, 0.3, -9.0, threads=4)
    assert f.tobytes() == rf.tobytes() and bits_equal(du, rdu) and bits_equal(dv, rdv)
    oracle.steps(p, f, b, 0.3, -9.0, du, dv, 60, threads=4)
    reference.steps(rf, rb, rbox, 0.3, -9.0, rdu, rdv, 60, threads=4)
    assert f.tobytes() == rf.tobytes() and bits_equal(du, rdu) and bits_equal(dv, rdv)


def test_scene_and_metaballs(oracle, reference):
    p = oracle.params()
    f, b = oracle.scene_default(p)
    rf, rb = reference.scene()
    assert f.tobytes() == rf.tobytes() and b.tobytes() == rb.tobytes()
    assert np.array_equal(oracle.metaballs(p, f), reference.metaballs(rf, reference.box()))
