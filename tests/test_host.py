"""Host logic and the C-ABI surface, CPU only: scene generators, gravity sources, exported symbols,
and the "no GPU => loud failure" contract."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, bits_equal, load_golden


def test_abi_library_loads_and_exports_every_declared_symbol(sph):
    L = C.CDLL(sph.LIB_HIP)
    header = open(os.path.join(ROOT, "include", "sph.h")).read()
    declared = set(re.findall(r"\b(sph_[a-z0-9_]+)\s*\(", header)) - {"sph_ctx"}
    assert declared == set(sph.ABI_SYMBOLS), declared ^ set(sph.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    # measurement / diagnostic entry points live in a header of their own (not part of the drop-in boundary)
    diag = open(os.path.join(ROOT, "include", "sph_diag.h")).read()
    ddecl = set(re.findall(r"\b(sph_[a-z0-9_]+)\s*\(", diag))
    assert ddecl == set(sph.DIAG_SYMBOLS), ddecl ^ set(sph.DIAG_SYMBOLS)
    assert not (ddecl & declared)
    for name in ddecl:
        assert hasattr(L, name), name
    H = C.CDLL(sph.LIB_HOST)
    hheader = open(os.path.join(ROOT, "include", "sph_host.h")).read()
    hdecl = set(re.findall(r"\b(sph_[a-z0-9_]+)\s*\(", hheader))
    for name in hdecl:
        assert hasattr(H, name), name
    assert L.sph_abi_version() == 8


def test_struct_layouts(sph):
    assert sph.PARTICLE.itemsize == 28                       # struct particle :26-31
    assert C.sizeof(sph.Params) == 18 * 4
    assert C.sizeof(sph.KernelTimes) == 8 * 4 + 4 + 4 + 4


def test_default_params_bit_patterns(sph):
    p = sph.default_params()
    g = load_golden("drop.npz")["constants"]
    got = np.array([p.r, p.h, p.x_max - p.x_min, p.y_max - p.y_min, p.rho0, p.c, p.g, p.dt, p.vol], np.float32)
    assert bits_equal(got, g[:9])
    hp = sph.Params()
    sph.hip_lib().sph_params_default(C.byref(hp))
    assert bytes(hp) == bytes(p)


def test_default_scene_matches_reference(sph):
    g = load_golden("drop.npz")
    prm, f, b = sph.scene("cfg0")
    assert len(f) == 269 and len(b) == 162
    assert bits_equal(np.stack([f["x"], f["y"]], 1), g["fluid_xy0"])
    assert bits_equal(np.stack([b["x"], b["y"]], 1), g["boundary_xy"])
    assert np.all(f["rho"] == 1000) and np.all(f["u"] == 0) and np.all(b["m"] == 0)
    assert bits_equal(f["m"], np.full(269, np.float32(prm.rho0) * np.float32(prm.vol), np.float32))


def test_block_scene_matches_fixture(sph):
    g = load_golden("block.npz")
    prm, f, b = sph.scene_block(tuple(g["box"]), 0.3, 0.3, 240, 60)
    assert bits_equal(np.stack([f["x"], f["y"]], 1), g["fluid_xy0"])
    assert bits_equal(np.stack([b["x"], b["y"]], 1), g["boundary_xy"])


def test_config_sizes(sph):
    """SURVEY.md §8d concrete configs."""
    prm, f, b = sph.scene("cfg1")
    assert len(f) == 262144 and len(b) == 16386
    # all inside the box, >= 4R from the walls, lattice spacing R
    assert f["x"].min() > 0.3 and f["y"].min() > 0.3 and f["y"].max() < 204.8
    prm, f, b = sph.scene_block((0.0, 120.0, 0.0, 60.0), 0.3, 0.3, 400, 500)
    assert len(f) == 200000 and abs(float(f["x"][500] - f["x"][0]) - 0.075) < 1e-6
    assert len(sph.dam_break(1)[2]) == 33600


def test_scene_capacity_errors(sph):
    p = sph.default_params()
    buf = np.zeros(10, sph.PARTICLE)
    L = sph.host_lib()
    assert L.sph_scene_default_fluid(C.byref(p), buf.ctypes.data_as(C.c_void_p), 10) == sph.SPH_E_ARG
    assert L.sph_scene_walls(C.byref(p), 1, buf.ctypes.data_as(C.c_void_p), 10) == sph.SPH_E_ARG
    assert L.sph_scene_block(C.byref(p), 0.3, 0.3, 4, 4, buf.ctypes.data_as(C.c_void_p), 10) == sph.SPH_E_ARG
    assert L.sph_scene_block(C.byref(p), 0.3, 0.3, -1, 4, None, 0) == sph.SPH_E_ARG


def test_gravity_sources(sph, tmp_path):
    gs = sph.GravitySource(sph.GRAVITY_CONSTANT, 9.81)
    assert gs.sample(0.0) == (0.0, pytest.approx(-9.81))                    # :442-443
    tilt = sph.GravitySource(sph.GRAVITY_TILT, 9.81)
    gx0, gy0 = tilt.sample(0.0)
    assert gx0 == pytest.approx(0.0, abs=1e-7) and gy0 == pytest.approx(-9.81)
    assert tilt.sample(0.05) == (gx0, gy0)                                  # zero-order hold, 10 Hz (:459)
    gx2, gy2 = tilt.sample(2.0)                                             # quarter period: theta = 15 deg
    assert gx2 == pytest.approx(9.81 * np.sin(np.radians(15)), rel=1e-6)
    assert gy2 == pytest.approx(-9.81 * np.cos(np.radians(15)), rel=1e-6)
    # MPU6050 sysfs reader with the axis mapping of :439-440
    (tmp_path / "in_accel_x_raw").write_text("8192\n")
    (tmp_path / "in_accel_y_raw").write_text("-16384\n")
    mpu = sph.GravitySource(sph.GRAVITY_MPU6050, 9.81, sysfs_dir=str(tmp_path))
    gx, gy = mpu.sample(0.0)
    assert gx == pytest.approx(-9.81) and gy == pytest.approx(-0.5 * 9.81)
    bad = sph.GravitySource(sph.GRAVITY_MPU6050, 9.81, sysfs_dir=str(tmp_path / "missing"))
    with pytest.raises(sph.SphError):
        bad.sample(0.0)


def test_create_argument_errors_and_no_cpu_fallback(sph):
    prm, f, b = sph.scene("cfg0")
    L = sph.hip_lib()
    h = C.c_void_p()
    assert L.sph_create(None, C.byref(prm), None, 0, None, 0, 0.0, -9.81, 0) == sph.SPH_E_ARG
    rc = L.sph_create(C.byref(h), C.byref(prm), None, 5, None, 0, 0.0, -9.81, 0)
    assert rc == sph.SPH_E_ARG and b"null" in L.sph_last_error(h)
    L.sph_destroy(h)
    f2 = f.copy()
    f2["m"][3] *= 2
    with pytest.raises(sph.SphError) as e:
        sph.Context(prm, f2, b)
    assert e.value.code == sph.SPH_E_ARG
    if L.sph_device_count() == 0:
        # the product path must fail loudly without a GPU — there is no CPU fallback
        with pytest.raises(sph.SphError) as e:
            sph.Context(prm, f, b)
        assert e.value.code == sph.SPH_E_HIP
    assert L.sph_error_string(sph.SPH_E_HIP).decode().startswith("HIP failure")
    assert L.sph_step(None, 0.0, 0.0, 1) == sph.SPH_E_ARG


def test_skin_is_a_per_context_parameter(sph):
    """the skin (neighbour-structure reuse) is a field of sph_params, fixed per context; slab hosts bin with the
    device cell 2H (1 + skin) of THEIR parameters (no process-wide state)."""
    L = sph.hip_lib()
    prm = sph.default_params()
    assert abs(prm.skin - 0.30) < 1e-7 and abs(prm.skin_min - 0.08) < 1e-7      # the largest and smallest skin
    two_h = np.float32(2) * np.float32(prm.h)
    for frac in (0.0, 0.1, 0.4):
        prm.skin = frac
        cell = np.float32(L.sph_device_cell(C.byref(prm)))
        assert abs(cell - two_h * (1 + np.float32(frac))) <= 2e-7
        assert sph.slab.device_cell(prm) == cell
        # the slab partitioner bins with that cell
        cols = sph.slab.grid_columns(prm)
        assert cols == int((np.float32(prm.x_max) - np.float32(prm.x_min)) / cell) + 1
    other = sph.default_params()                      # untouched by the loop above
    assert abs(np.float32(L.sph_device_cell(C.byref(other))) - two_h * np.float32(1.30)) <= 2e-7
    prm.skin = 1.5
    assert L.sph_device_cell(C.byref(prm)) == 0.0      # invalid
    _, f, b = sph.scene("cfg0")
    with pytest.raises(sph.SphError) as e:
        sph.Context(prm, f, b)
    assert e.value.code == sph.SPH_E_ARG               # rejected before any device is touched


def test_multi_layer_wall_generator(sph):
    prm = sph.default_params((0.0, 6.0, 0.0, 4.0))
    inner = (0.5, 5.5, 0.5, 3.5)
    one, three = sph.scene_walls_layers(prm, inner, 1), sph.scene_walls_layers(prm, inner, 3)
    r = np.float32(prm.r)
    for w, layers in ((one, 1), (three, 3)):
        xy = np.stack([w["x"], w["y"]], 1)
        assert len(np.unique(xy, axis=0)) == len(w)                      # corners sampled once
        assert abs(w["x"].min() - (0.5 - (layers - 1) * r)) < 1e-6 and abs(w["y"].max() - (3.5 + (layers - 1) * r)) < 1e-6
        on_frame = (np.isclose(w["x"], w["x"].min()) | np.isclose(w["x"], w["x"].max()) |
                    np.isclose(w["y"], w["y"].min()) | np.isclose(w["y"], w["y"].max()))
        assert on_frame.sum() >= len(w) // layers - 8
        assert np.all(w["m"] == 0) and np.all(w["rho"] == prm.rho0)
    assert len(three) > 3 * len(one)                                     # outer frames are longer
    L = sph.host_lib()
    assert L.sph_scene_walls_layers(C.byref(prm), 0.05, 5.5, 0.5, 3.5, 2, None, 0) == sph.SPH_E_ARG      # outer frame leaves the box


def test_wall_motion_from_accelerometer(sph):
    """sph_wall_motion (README.md:175-176): a constant tilt gives no velocity; a jolt of the box shows up as a velocity
    of the right sign and size, which leaks away afterwards; the numbers follow the documented recurrence."""
    wm = sph.WallMotion()
    dt = 1e-3
    for _ in range(2000):                                  # the box at rest, tilted by 10 degrees
        v = wm.update(9.81 * np.sin(np.radians(10)), -9.81 * np.cos(np.radians(10)), dt)
    assert abs(v[0]) < 1e-6 and abs(v[1]) < 1e-6
    gx0, gy0 = 9.81 * np.sin(np.radians(10)), -9.81 * np.cos(np.radians(10))
    # a 50 ms jolt: the box accelerates at +4 m/s^2 along x, so apparent gravity gains -4 m/s^2 along x
    ref = dict(glx=np.float32(gx0), gly=np.float32(gy0), vx=np.float32(v[0]), vy=np.float32(v[1]))
    for k in range(50):
        gx = gx0 - 4.0
        v = wm.update(gx, gy0, dt)
        kk = np.float32(min(dt / 0.5, 1.0))
        ref["glx"] = np.float32(ref["glx"] + (np.float32(gx) - ref["glx"]) * kk)
        ax = np.float32(-(np.float32(gx) - ref["glx"]))
        ref["vx"] = np.float32((ref["vx"] + ax * np.float32(dt)) * np.float32(np.exp(np.float32(-dt / 1.0))))
    assert v[0] == pytest.approx(float(ref["vx"]), rel=1e-4)
    assert 0.15 < v[0] < 0.2 and abs(v[1]) < 1e-6           # ~ 4 m/s^2 x 50 ms = 0.2 m/s, less what the low pass took for tilt
    peak = v[0]
    for _ in range(3000):                                  # at rest again: the velocity leaks away (and the filter's overshoot with it)
        v = wm.update(gx0, gy0, dt)
    assert abs(v[0]) < 0.1 * peak
