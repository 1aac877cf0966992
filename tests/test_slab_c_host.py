"""The C multi-GPU host (pi-sph-fluid_amd/host/slab_sph_fluid.c: one process per GPU, RCCL over xGMI) and the C
partitioner it uses (host/sph_slab_host.c).

CPU: the C partitioner equals the Python one (pi-sph-fluid_amd/slab.py, itself pinned by the gloo tests), a rank
generates exactly the lattice columns the Python host would give it, and `slab_sph_fluid --ranks 2` started without a
GPU fails loudly and at once (no CPU path, no rendezvous hang).
GPU (one MI355X): the one-rank run through ncclCommInitRank / the slab entry points equals sph_step on a single
context (--check), conserves every particle on the 2M-particle scene, and asking for more ranks than GPUs is refused
with a message instead of a hang."""
import ctypes as C
import json
import os
import subprocess
import time

import numpy as np
import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "pi-sph-fluid_amd", "host", "slab_sph_fluid")


def test_c_partitioner_equals_python(sph):
    L = sph.host_lib()
    for spec, worlds in ((sph.dam_break_spec(1), (1, 2, 3, 8)), (sph.BLOCK_SCENES["cfg4"], (1, 4, 8)),
                         (((0.0, 40.0, 0.0, 8.0), 0.3, 0.3, 240, 60), (1, 2, 5))):
        box, x0, y0, nx, ny = spec
        for skin in (0.15, 0.0, 0.3):
            prm = sph.default_params(box, skin)
            assert L.sph_slab_grid_columns(C.byref(prm)) == sph.slab.grid_columns(prm)
            for world in worlds:
                cuts = (C.c_int * (world + 1))()
                assert L.sph_slab_partition_block(C.byref(prm), x0, nx, ny, world, cuts) == 0
                py = sph.slab.partition_block(prm, spec, world)
                assert [(cuts[r], cuts[r + 1]) for r in range(world)] == py
                c0, c1 = py[world // 2]
                ib, ie = C.c_long(), C.c_long()
                assert L.sph_slab_block_columns(C.byref(prm), x0, nx, c0, c1, C.byref(ib), C.byref(ie)) == 0
                gc = sph.slab.block_lattice_columns(prm, spec)
                sel = np.nonzero((gc >= c0 - 2) & (gc < c1 + 2))[0]
                assert (ib.value, ie.value) == (int(sel[0]), int(sel[-1]) + 1)
    prm = sph.default_params((0.0, 4.0, 0.0, 2.0))
    cuts = (C.c_int * 9)()
    assert L.sph_slab_partition_block(C.byref(prm), 0.3, 20, 10, 8, cuts) == sph.SPH_E_ARG      # too narrow for 8 slabs


def test_c_host_fails_loudly_without_gpu(sph):
    if sph.hip_lib().sph_device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu tests below")
    t0 = time.time()
    r = subprocess.run([HOST, "--ranks", "2", "--block", "400", "100", "60", "20", "--steps", "3", "--warmup", "1"],
                       capture_output=True, timeout=120)
    assert r.returncode != 0
    assert b"no HIP device available" in r.stderr and b"no CPU path" in r.stderr
    assert not r.stdout.strip()
    assert time.time() - t0 < 60


@pytest.mark.gpu
def test_c_host_one_rank_equals_sph_step(sph):
    r = subprocess.run([HOST, "--ranks", "1", "--block", "600", "150", "90", "20", "--steps", "150", "--warmup", "50", "--check"],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode().splitlines()
    rec = json.loads([ln for ln in out if ln.startswith("{")][0])      # (RCCL prints a version banner on stdout first)
    assert rec["n_gpus"] == 1 and rec["n_fluid"] == 90000 and rec["particles_conserved"] is True
    assert rec["ticks_per_s"] > 0 and 0 < rec["neighbour_rebuilds"] < 200
    chk = [ln for ln in out if ln.startswith("check:")]
    assert len(chk) == 1 and chk[0].endswith("-> ok"), chk


@pytest.mark.gpu
def test_c_host_tilt_run_and_rank_count_guard(sph):
    r = subprocess.run([HOST, "--ranks", "1", "--scene", "dam", "--steps", "60", "--warmup", "20", "--tilt"],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert rec["n_fluid"] == 2000000 and rec["particles_conserved"] is True and "tilt" in rec["workload"]
    ndev = sph.hip_lib().sph_device_count()
    r = subprocess.run([HOST, "--ranks", str(ndev + 1), "--block", "400", "100", "60", "20", "--steps", "2", "--warmup", "1"],
                       capture_output=True, timeout=120)
    assert r.returncode != 0 and b"RCCL does not share a device" in r.stderr


@pytest.mark.gpu
def test_c_host_two_ranks_over_rccl(sph):
    """two ranks on two GPUs (skipped on a one-GPU box: RCCL does not share a device): the halo exchange and the reduction of
    the rebuild word over RCCL; every particle owned exactly once at the end, rebuilds in step on both ranks."""
    if sph.hip_lib().sph_device_count() < 2:
        pytest.skip("needs two GPUs")
    r = subprocess.run([HOST, "--ranks", "2", "--block", "600", "150", "90", "20", "--steps", "300", "--warmup", "100"],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["n_fluid"] == 90000 and rec["particles_conserved"] is True
    assert rec["ticks_per_s"] > 0 and 0 < rec["neighbour_rebuilds"] < 400


@pytest.mark.gpu
def test_c_host_deterministic_equals_sph_step_bitwise(sph):
    """--deterministic (sph_params.deterministic): the slab path through the C host and sph_step give the same bits when
    they rebuild in the same steps (skin 0: every step; with a skin the slab path may rebuild a step earlier)"""
    r = subprocess.run([HOST, "--ranks", "1", "--block", "600", "150", "90", "20", "--steps", "150", "--warmup", "50", "--check",
                        "--deterministic", "--skin", "0"], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    chk = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("check:")]
    assert len(chk) == 1 and "max|dx| = 0.000e+00" in chk[0] and "max|drho| = 0.000e+00" in chk[0], chk
