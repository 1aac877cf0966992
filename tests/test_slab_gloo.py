"""The N > 1 path on CPU: world_size-2 (and 3) `gloo` runs of the product's host-side slab logic
(partition_columns, halo protocol, TorchTransport, SlabRunner) with the oracle as the per-slab compute.
Merged results must equal the single-rank oracle BIT-EXACTLY (slab-independent summation order), which pins:
two ghost columns suffice for one exchange per step, migrants change owner exactly once, nothing is lost."""
import multiprocessing as mp
import socket

import numpy as np
import pytest

from conftest import bits_equal
import slab_oracle


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, scene, nsteps, rebalance_at=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=slab_oracle.gloo_worker, args=(r, world, port, scene, nsteps, q, rebalance_at)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@pytest.mark.parametrize("world,scene,nsteps", [(2, "block_moving", 40), (3, "block_moving", 25), (2, "drop", 30)])
def test_slabs_equal_single_rank_bit_exact(sph, orc, oracle, world, scene, nsteps):
    prm, f, b = slab_oracle.scene_build(sph, scene)
    p = oracle.params((prm.x_min, prm.x_max, prm.y_min, prm.y_max))
    of, ob = f.view(orc.PARTICLE).copy(), b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    du, dv = oracle.eval(p, of, ob, 0.0, -9.81, threads=2)
    oracle.steps(p, of, ob, 0.0, -9.81, du, dv, nsteps, threads=2)

    res = run_world(world, scene, nsteps)
    seen = np.zeros(len(f), int)
    migrated = 0
    for rank, own, ids, sdu, sdv, mig, (c0, c1) in res:
        seen[ids] += 1
        migrated += mig
        for fld in ("x", "y", "u", "v", "rho", "p"):
            assert bits_equal(own[fld], of[fld][ids]), (rank, fld)
        assert bits_equal(sdu, du[ids]) and bits_equal(sdv, dv[ids])
        gc = sph.slab.global_columns(prm, own["x"])
        assert np.all((gc >= c0) & (gc < c1))            # ownership follows position
    assert np.all(seen == 1)                              # every particle owned exactly once
    if scene == "block_moving":
        assert migrated > 0                               # the test really exercised migration


@pytest.mark.parametrize("world", [2, 3])
def test_rebalance_over_gloo(sph, orc, oracle, world):
    """dynamic re-balancing with one slab per rank: after 70 steps of a block flying sideways (two cell columns) the ranks
    re-partition from the current column histogram, ship every particle to the slab that now holds its column and
    continue; the merged result is the single-rank oracle's to rounding (a re-created slab re-evaluates a from (x, v),
    the reference's init sequence: continued, not bit-continued), nothing lost or duplicated, ownership by position."""
    nsteps, at = 100, 70
    prm, f, b = slab_oracle.scene_build(sph, "block_moving")
    p = oracle.params((prm.x_min, prm.x_max, prm.y_min, prm.y_max))
    of, ob = f.view(orc.PARTICLE).copy(), b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    du, dv = oracle.eval(p, of, ob, 0.0, -9.81, threads=2)
    oracle.steps(p, of, ob, 0.0, -9.81, du, dv, nsteps, threads=2)
    res = run_world(world, "block_moving", nsteps, rebalance_at=at)
    seen = np.zeros(len(f), int)
    ranges = []
    for rank, own, ids, sdu, sdv, mig, (c0, c1) in res:
        seen[ids] += 1
        ranges.append((c0, c1))
        assert max(np.abs(own["x"] - of["x"][ids]).max(), np.abs(own["y"] - of["y"][ids]).max()) <= 5e-6
        assert np.max(np.abs(own["rho"] - of["rho"][ids]) / of["rho"][ids]) <= 1e-5
        gc = sph.slab.global_columns(prm, own["x"])
        assert np.all((gc >= c0) & (gc < c1))
    assert np.all(seen == 1)
    assert ranges[0][0] == 0 and all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
    first = sph.slab.partition_columns(prm, f, world, slack=8)
    assert [z for _, z in ranges[:-1]] != [z for _, z in first[:-1]]            # the cuts followed the block
    assert ranges[-1][1] == sph.slab.grid_columns(prm)                           # and the slabs now tile the whole box
    counts = [len(r[1]) for r in res]
    assert max(counts) - min(counts) <= 0.25 * len(f) / world                    # balanced again (30 steps later)


def test_partition_and_halo_format(sph):
    prm, f, b = sph.dam_break(1)
    for world in (1, 2, 4, 8):
        parts = sph.slab.partition_columns(prm, f, world)
        assert parts[0][0] == 0 and all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
        gc = sph.slab.global_columns(prm, f["x"])
        counts = [int(((gc >= a) & (gc < z)).sum()) for a, z in parts]
        assert sum(counts) == len(f) and max(counts) - min(counts) <= 0.02 * len(f) / world + 1500
        assert all(z - a >= 4 for a, z in parts)
    # the slabs tile the whole box (a dam-break front never leaves the decomposition) ...
    assert parts[0][0] == 0 and parts[-1][1] == sph.slab.grid_columns(prm)
    # ... unless a slack is given (small local grids): then the dry part beyond it belongs to no slab
    tight = sph.slab.partition_columns(prm, f, 4, slack=64)
    assert tight[-1][1] < sph.slab.grid_columns(prm)
    # a lattice block is partitioned without generating it, and a slab host generates only its own columns
    spec = sph.dam_break_spec(1)
    assert sph.slab.partition_block(prm, spec, 8) == parts
    c0, c1 = parts[3]
    loc, ids = sph.slab.local_block_subset(sph, prm, spec, c0, c1)
    ref_loc, ref_ids = sph.slab.local_subset(prm, f, c0, c1)
    assert np.array_equal(ids, ref_ids) and loc.tobytes() == ref_loc.tobytes()
    assert sph.slab.halo_words(10) == 4 + 5 * 10
    with pytest.raises(ValueError):
        sph.slab.partition_columns(*sph.scene("cfg0")[:2], 8)
