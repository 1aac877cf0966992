"""pi-sph-fluid_amd — MI355X-native 2-D WCSPH stepper (host-side Python binding).

The product is the C-ABI library ``csrc/libsph_hip.so`` (hand-written gfx950 kernels,
include/sph.h) plus the plain-C host helpers ``host/libsph_host.so`` (include/sph_host.h).
This module is a thin ctypes mirror of those two headers, used by tests/ and bench.py; the
C host program lives in host/desktop_sph_fluid.c.  The directory name has a hyphen, so import
it with ``importlib.import_module("pi-sph-fluid_amd")``.

There is no CPU compute path here: creating a :class:`Context` without a gfx950 GPU raises
:class:`SphError` (SPH_E_HIP).  The CPU oracle under ``oracle/`` is test infrastructure and is
never imported from this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_HIP = os.path.join(HERE, "csrc", "libsph_hip.so")      # (measurement scripts may point this at an A/B build before first use)
LIB_HOST = os.environ.get("SPH_HOST_LIB") or os.path.join(HERE, "host", "libsph_host.so")      # (SPH_HOST_LIB: the sanitizer build, tests/test_sanitizers.py)

# byte-compatible with the reference's `struct particle` (pi_sph_fluid.c:26-31)
PARTICLE = np.dtype([("x", "<f4"), ("y", "<f4"), ("u", "<f4"), ("v", "<f4"),
                     ("m", "<f4"), ("rho", "<f4"), ("p", "<f4")])

SPH_OK, SPH_E_ARG, SPH_E_HIP, SPH_E_OUT_OF_DOMAIN, SPH_E_NAN = 0, -1, -2, -3, -4
SPH_E_NOMEM, SPH_E_CAPACITY, SPH_E_STATE = -5, -6, -7
KERNEL_NAMES = ["kick_drift", "key_hist", "scan", "reorder", "build_list", "density_eos", "force_kick", "halo"]
# algorithmic HBM bytes per fluid particle per launch (P1 44 incl. 4 B key, P2 5.2, P4 44, P5 16.6,
# P6 40).  The single-GPU step's force launch also does the next step's kick 1/2 + drift (P1 without the key): 40 + 40.
KERNEL_ALGO_BYTES = {"kick_drift": 40.0, "key_hist": 4.0 + 5.2, "reorder": 44.0, "density_eos": 16.6, "force_kick": 40.0 + 40.0}
STEP_ALGO_BYTES = 152.0

# every symbol include/sph.h and include/sph_host.h declare
ABI_SYMBOLS = [      # include/sph.h: the drop-in boundary (+ slabs, metaballs)
    "sph_params_default", "sph_abi_version", "sph_error_string", "sph_device_count",
    "sph_create", "sph_destroy", "sph_last_error", "sph_step", "sph_sync",
    "sph_read_particles", "sph_read_accel", "sph_read_boundary", "sph_update_boundary", "sph_set_boundary_velocity", "sph_stats",
    "sph_n_fluid", "sph_n_boundary", "sph_grid_dims", "sph_device_grid", "sph_out_of_domain_count",
    "sph_device_cell", "sph_request_rebuild", "sph_set_rebuild_launches", "sph_get_rebuild_launches", "sph_rebuild_stats",
    "sph_current_skin", "sph_set_verification", "sph_set_list_repair",
    "sph_upload_state", "sph_upload_accel", "sph_eval_density", "sph_eval_pressure", "sph_eval_accel",
    "sph_set_stream", "sph_device_bytes",
    "sph_render_metaballs",
    "sph_create_slab", "sph_slab_step_begin", "sph_slab_step_pack", "sph_slab_step_overlap", "sph_slab_step_overlap_on", "sph_slab_step_end",
    "sph_slab_peer_reduce", "sph_slab_peer_push", "sph_slab_peer_wait", "sph_slab_set_peer_links", "sph_slab_step", "sph_slab_steps", "sph_slab_set_speculative",
    "sph_slab_flag_buffer", "sph_slab_set_flag_buffer", "sph_slab_flag_get", "sph_slab_flag_set", "sph_slab_buffers", "sph_slab_set_buffers",
    "sph_slab_copy_out", "sph_slab_copy_in", "sph_slab_read", "sph_slab_counts", "sph_slab_halo_bytes",
]
DIAG_SYMBOLS = [     # include/sph_diag.h: measurement and diagnostics (bench.py, profiling, tests)
    "sph_profile_steps", "sph_time_kernel", "sph_set_variant",
    "sph_direct_tile_reasons", "sph_verify_stats", "sph_rebuild_reasons", "sph_check_stats", "sph_repair_stats",
    "sph_box_calibrate",
]
HOST_SYMBOLS = [
    "sph_params_default", "sph_scene_default_fluid", "sph_scene_walls", "sph_scene_disc", "sph_scene_block",
    "sph_scene_block_range", "sph_scene_walls_layers",
    "sph_gravity_init", "sph_gravity_sample", "sph_wall_motion_init", "sph_wall_motion_update",
    "sph_slab_grid_columns", "sph_slab_column_of", "sph_slab_partition_block", "sph_slab_partition_counts", "sph_slab_block_columns",
]


class Params(C.Structure):
    """sph_params of include/sph.h (the reference's #defines :11-20 + the box of :595)."""
    _fields_ = [("r", C.c_float), ("h", C.c_float), ("rho0", C.c_float), ("c", C.c_float),
                ("g", C.c_float), ("dt", C.c_float), ("vol", C.c_float),
                ("x_min", C.c_float), ("x_max", C.c_float), ("y_min", C.c_float), ("y_max", C.c_float),
                ("alpha", C.c_float), ("eps", C.c_float), ("k1", C.c_float), ("k2", C.c_float),
                ("skin", C.c_float), ("deterministic", C.c_int), ("skin_min", C.c_float)]


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_float * 8), ("step_ms", C.c_float), ("nsteps", C.c_int), ("rebuilds", C.c_int)]


class BoxCalibration(C.Structure):
    """sph_box_calibration of include/sph_diag.h."""
    _fields_ = [("copy_gbs", C.c_float), ("valu_cycles", C.c_float), ("clock_ghz", C.c_float), ("copy_ms", C.c_float), ("valu_ms", C.c_float)]


class SlabDesc(C.Structure):
    """sph_slab_desc of include/sph.h."""
    _fields_ = [("col_begin", C.c_int), ("col_end", C.c_int), ("has_left", C.c_int), ("has_right", C.c_int),
                ("halo_capacity", C.c_int), ("particle_capacity", C.c_int)]


class Gravity(C.Structure):
    _fields_ = [("kind", C.c_int), ("g", C.c_float), ("amp_deg", C.c_float), ("period_s", C.c_float),
                ("hold_s", C.c_float), ("sysfs_dir", C.c_char * 256),
                ("last_t", C.c_float), ("gx", C.c_float), ("gy", C.c_float), ("primed", C.c_int)]


GRAVITY_CONSTANT, GRAVITY_TILT, GRAVITY_MPU6050 = 0, 1, 2


class WallMotionState(C.Structure):
    _fields_ = [("tau_tilt", C.c_float), ("tau_leak", C.c_float), ("glx", C.c_float), ("gly", C.c_float),
                ("vx", C.c_float), ("vy", C.c_float), ("primed", C.c_int)]


class SphError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("sph error %d: %s" % (code, message))
        self.code = code


def build(verbose=False):
    """Compile every native piece for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", HERE, "all"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


_hip = None
_host = None


def hip_lib():
    """The C-ABI library; raises loudly when it is not built (no fallback)."""
    global _hip
    if _hip is None:
        if not os.path.exists(LIB_HIP):
            raise SphError(SPH_E_HIP, "%s not built — run `make -C %s` (there is no CPU fallback)" % (LIB_HIP, HERE))
        L = C.CDLL(LIB_HIP)
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.sph_params_default.argtypes = [C.POINTER(Params)]
        L.sph_error_string.restype = C.c_char_p
        L.sph_error_string.argtypes = [ci]
        L.sph_create.argtypes = [C.POINTER(vp), C.POINTER(Params), vp, ci, vp, ci, cf, cf, ci]
        L.sph_destroy.argtypes = [vp]
        L.sph_destroy.restype = None
        L.sph_last_error.argtypes = [vp]
        L.sph_last_error.restype = C.c_char_p
        L.sph_step.argtypes = [vp, cf, cf, ci]
        L.sph_sync.argtypes = [vp]
        L.sph_read_particles.argtypes = [vp, vp]
        L.sph_read_accel.argtypes = [vp, vp, vp]
        L.sph_read_boundary.argtypes = [vp, vp]
        L.sph_update_boundary.argtypes = [vp, vp]
        L.sph_set_boundary_velocity.argtypes = [vp, C.c_float, C.c_float]
        L.sph_stats.argtypes = [vp, C.POINTER(cf), C.POINTER(cf)]
        L.sph_n_fluid.argtypes = [vp]
        L.sph_n_boundary.argtypes = [vp]
        L.sph_grid_dims.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
        L.sph_out_of_domain_count.argtypes = [vp]
        L.sph_out_of_domain_count.restype = C.c_longlong
        L.sph_device_grid.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(cf)]
        L.sph_device_cell.argtypes = [C.POINTER(Params)]
        L.sph_device_cell.restype = cf
        L.sph_request_rebuild.argtypes = [vp]
        L.sph_set_rebuild_launches.argtypes = [vp, C.c_int]
        L.sph_get_rebuild_launches.argtypes = [vp]
        L.sph_rebuild_stats.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.sph_check_stats.argtypes = [vp, C.POINTER(C.c_longlong)]
        L.sph_verify_stats.argtypes = [vp, C.POINTER(C.c_longlong)]
        L.sph_rebuild_reasons.argtypes = [vp, C.POINTER(C.c_longlong)]
        L.sph_direct_tile_reasons.argtypes = [vp, C.POINTER(C.c_longlong)]
        L.sph_current_skin.argtypes = [vp]
        L.sph_current_skin.restype = C.c_float
        L.sph_upload_state.argtypes = [vp, vp]
        L.sph_upload_accel.argtypes = [vp, vp, vp]
        L.sph_eval_density.argtypes = [vp]
        L.sph_eval_pressure.argtypes = [vp]
        L.sph_eval_accel.argtypes = [vp, cf, cf]
        L.sph_profile_steps.argtypes = [vp, cf, cf, ci, C.POINTER(KernelTimes)]
        L.sph_time_kernel.argtypes = [vp, ci, ci, C.POINTER(cf)]
        L.sph_set_stream.argtypes = [vp, vp]
        L.sph_device_bytes.argtypes = [vp]
        L.sph_device_bytes.restype = C.c_size_t
        L.sph_set_variant.argtypes = [vp, ci]
        L.sph_render_metaballs.argtypes = [vp, vp]
        L.sph_slab_halo_bytes.argtypes = [C.POINTER(Params), ci]
        L.sph_slab_halo_bytes.restype = C.c_size_t
        L.sph_create_slab.argtypes = [C.POINTER(vp), C.POINTER(Params), C.POINTER(SlabDesc), vp, vp, ci, vp, ci, cf, cf, ci]
        L.sph_slab_step_begin.argtypes = [vp, cf, cf]
        L.sph_slab_step_pack.argtypes = [vp]
        L.sph_slab_step_overlap.argtypes = [vp]
        L.sph_slab_step_overlap_on.argtypes = [vp, vp]
        L.sph_slab_step_end.argtypes = [vp]
        L.sph_slab_flag_buffer.argtypes = [vp, C.POINTER(vp)]
        L.sph_slab_set_flag_buffer.argtypes = [vp, vp]
        L.sph_slab_flag_get.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.sph_slab_flag_set.argtypes = [vp, C.c_uint32]
        L.sph_slab_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.sph_slab_set_buffers.argtypes = [vp, vp, vp, vp, vp, C.c_size_t]
        L.sph_slab_copy_out.argtypes = [vp, ci, vp]
        L.sph_slab_copy_in.argtypes = [vp, ci, vp]
        L.sph_slab_read.argtypes = [vp, vp, vp, vp, vp, ci, C.POINTER(ci)]
        L.sph_slab_counts.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(LIB_HOST):
            raise SphError(SPH_E_ARG, "%s not built — run `make -C %s`" % (LIB_HOST, HERE))
        L = C.CDLL(LIB_HOST)
        vp, cl, cf = C.c_void_p, C.c_long, C.c_float
        L.sph_params_default.argtypes = [C.POINTER(Params)]
        L.sph_scene_default_fluid.argtypes = [C.POINTER(Params), vp, cl]
        L.sph_scene_default_fluid.restype = cl
        L.sph_scene_walls.argtypes = [C.POINTER(Params), C.c_int, vp, cl]
        L.sph_scene_walls.restype = cl
        L.sph_scene_disc.argtypes = [C.POINTER(Params), cf, cf, cf, vp, cl]
        L.sph_scene_disc.restype = cl
        L.sph_scene_block.argtypes = [C.POINTER(Params), cf, cf, cl, cl, vp, cl]
        L.sph_scene_block.restype = cl
        L.sph_scene_walls_layers.argtypes = [C.POINTER(Params), cf, cf, cf, cf, C.c_int, vp, cl]
        L.sph_scene_walls_layers.restype = cl
        L.sph_scene_block_range.argtypes = [C.POINTER(Params), cf, cf, cl, cl, cl, cl, vp, cl]
        L.sph_scene_block_range.restype = cl
        L.sph_slab_grid_columns.argtypes = [C.POINTER(Params)]
        L.sph_slab_column_of.argtypes = [C.POINTER(Params), cf]
        L.sph_slab_partition_block.argtypes = [C.POINTER(Params), cf, cl, cl, C.c_int, C.POINTER(C.c_int)]
        L.sph_slab_block_columns.argtypes = [C.POINTER(Params), cf, cl, C.c_int, C.c_int, C.POINTER(cl), C.POINTER(cl)]
        L.sph_slab_partition_counts.argtypes = [C.POINTER(C.c_longlong), C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.sph_gravity_init.argtypes = [C.POINTER(Gravity), C.c_int, cf]
        L.sph_gravity_init.restype = None
        L.sph_gravity_sample.argtypes = [C.POINTER(Gravity), cf, C.POINTER(cf), C.POINTER(cf)]
        L.sph_wall_motion_init.argtypes = [C.POINTER(WallMotionState)]
        L.sph_wall_motion_init.restype = None
        L.sph_wall_motion_update.argtypes = [C.POINTER(WallMotionState), cf, cf, cf, C.POINTER(cf), C.POINTER(cf)]
        L.sph_wall_motion_update.restype = None
        _host = L
    return _host


def default_params(box=None, skin=None, deterministic=False):
    """reference defaults (:11-20); box = (x_min, x_max, y_min, y_max); skin = Verlet skin as a fraction of 2H
    (None: the library default, a skin that adapts between skin_min and skin; a number: that skin, fixed; 0: rebuild
    the neighbour structure every step like the reference, :626)."""
    p = Params()
    host_lib().sph_params_default(C.byref(p))
    if box is not None:
        p.x_min, p.x_max, p.y_min, p.y_max = [float(v) for v in box]
    if skin is not None:
        p.skin = p.skin_min = float(skin)
    p.deterministic = 1 if deterministic else 0
    return p


def _two_call(fn, *args):
    n = fn(*args, None, 0)
    if n < 0:
        raise SphError(int(n), "scene generator rejected its arguments")
    out = np.zeros(n, PARTICLE)
    m = fn(*args, out.ctypes.data_as(C.c_void_p), n)
    if m != n:
        raise SphError(int(m), "scene generator size mismatch")
    return out


# ---- scenes ----
def scene_default(prm=None):
    """cfg0: the exact default scene of the reference (269 fluid + 162 boundary)."""
    prm = prm or default_params()
    L = host_lib()
    return prm, _two_call(L.sph_scene_default_fluid, C.byref(prm)), _two_call(L.sph_scene_walls, C.byref(prm), 1)


def scene_walls(prm, accumulate=False):
    """the wall generator of :523-540 for the box of prm (single layer, spacing R)."""
    return _two_call(host_lib().sph_scene_walls, C.byref(prm), 1 if accumulate else 0)


def scene_walls_layers(prm, wall_box, layers):
    """`layers` nested frames of wall particles, the innermost on wall_box = (x0, x1, y0, y1), all inside prm's box."""
    return _two_call(host_lib().sph_scene_walls_layers, C.byref(prm), *[float(v) for v in wall_box], int(layers))


def scene_disc(box, cx, cy, radius):
    prm = default_params(box)
    L = host_lib()
    return prm, _two_call(L.sph_scene_disc, C.byref(prm), cx, cy, radius), _two_call(L.sph_scene_walls, C.byref(prm), 0)


def scene_block(box, x0, y0, nx, ny):
    prm = default_params(box)
    L = host_lib()
    return prm, _two_call(L.sph_scene_block, C.byref(prm), x0, y0, nx, ny), _two_call(L.sph_scene_walls, C.byref(prm), 0)


def block_range(prm, x0, y0, nx, ny, i_begin, i_end):
    """lattice columns [i_begin, i_end) of the nx x ny block (bit-identical to that part of scene_block)."""
    return _two_call(host_lib().sph_scene_block_range, C.byref(prm), x0, y0, nx, ny, i_begin, i_end)


# the dam-break family as (box, x0, y0, nx, ny): what a slab host needs to generate only its own columns
BLOCK_SCENES = {
    "cfg2": ((0.0, 1200.0, 0.0, 60.0), 0.3, 0.3, 4000, 500),
    "cfg3": ((0.0, 2400.0, 0.0, 60.0), 0.3, 0.3, 16000, 500),
    "cfg4": ((0.0, 2400.6, 0.0, 150.0), 0.3, 0.3, 32000, 1000),
    "cfg4_slab": ((0.0, 300.6, 0.0, 150.0), 0.3, 0.3, 4000, 1000),      # one eighth of cfg4 as a tank of its own: what one GPU of the 8-GPU run holds
}


def dam_break_spec(n_slabs):
    return ((0.0, 1200.0 * n_slabs, 0.0, 60.0), 0.3, 0.3, 4000 * n_slabs, 500)


def scene(name):
    """Named configurations of BASELINE.json / SURVEY.md §8d."""
    if name == "cfg0":
        return scene_default()
    if name == "cfg1":      # 256k-particle drop on a dry surface
        return scene_disc((0.0, 409.6, 0.0, 204.8), 204.8, 30.95, 21.665)
    if name == "cfg2":      # 2M dam break
        return scene_block((0.0, 1200.0, 0.0, 60.0), 0.3, 0.3, 4000, 500)
    if name == "cfg3":      # 8M dam break (4 x-slabs)
        return scene_block((0.0, 2400.0, 0.0, 60.0), 0.3, 0.3, 16000, 500)
    if name == "cfg4":      # 32M tank (8 x-slabs)
        return scene_block((0.0, 2400.6, 0.0, 150.0), 0.3, 0.3, 32000, 1000)
    if name == "cfg4_slab":      # 4M: the particles of one of cfg4's eight slabs (bench.py: the slab step's overhead at that size)
        return scene_block((0.0, 300.6, 0.0, 150.0), 0.3, 0.3, 4000, 1000)
    raise ValueError("unknown scene %r" % name)


def dam_break(n_slabs):
    """The cfg2 -> cfg3 family: one 4000 x 500 block (2M particles, 1200 m of box) per slab."""
    return scene_block((0.0, 1200.0 * n_slabs, 0.0, 60.0), 0.3, 0.3, 4000 * n_slabs, 500)


class GravitySource:
    """get_gravity / get_gravity_routine of the reference (:431-464) behind one sample(t) call."""

    def __init__(self, kind=GRAVITY_CONSTANT, g=9.81, **kw):
        self.s = Gravity()
        host_lib().sph_gravity_init(C.byref(self.s), kind, g)
        for k, v in kw.items():
            setattr(self.s, k, v.encode() if k == "sysfs_dir" else v)

    def sample(self, t):
        gx, gy = C.c_float(), C.c_float()
        rc = host_lib().sph_gravity_sample(C.byref(self.s), t, C.byref(gx), C.byref(gy))
        if rc:
            raise SphError(rc, "gravity source failed")
        return gx.value, gy.value


class WallMotion:
    """sph_wall_motion (include/sph_host.h): the velocity of the box inferred from the accelerometer's gravity samples
    (README.md:175-176), to be handed to Context.set_boundary_velocity."""

    def __init__(self, tau_tilt=None, tau_leak=None):
        self.s = WallMotionState()
        host_lib().sph_wall_motion_init(C.byref(self.s))
        if tau_tilt is not None:
            self.s.tau_tilt = tau_tilt
        if tau_leak is not None:
            self.s.tau_leak = tau_leak

    def update(self, gx, gy, dt):
        vx, vy = C.c_float(), C.c_float()
        host_lib().sph_wall_motion_update(C.byref(self.s), gx, gy, dt, C.byref(vx), C.byref(vy))
        return vx.value, vy.value


def box_calibrate(device=0, repeats=3):
    """sph_box_calibrate (include/sph_diag.h), `repeats` times: the median of each figure — what THIS box delivers to a streaming copy
    and to a saturated v_fma_f32 stream, so that a steps/s figure can be told from the box it was measured on."""
    L = hip_lib()
    L.sph_box_calibrate.argtypes = [C.c_int, C.POINTER(BoxCalibration)]
    rows = []
    for _ in range(repeats):
        b = BoxCalibration()
        rc = L.sph_box_calibrate(device, C.byref(b))
        if rc:
            raise SphError(rc, "sph_box_calibrate failed")
        rows.append((b.copy_gbs, b.valu_cycles, b.clock_ghz))
    med = [float(np.median([r[k] for r in rows])) for k in range(3)]
    return {"copy_gbs": round(med[0], 1), "valu_cycles": round(med[1], 3), "clock_ghz": round(med[2], 3) if med[2] > 0 else None,
            "repeats": repeats, "copy_gbs_runs": [round(r[0], 1) for r in rows], "valu_cycles_runs": [round(r[1], 3) for r in rows]}


class Context:
    """One sph_ctx: the init/step/read-back surface of include/sph.h."""

    def __init__(self, prm, fluid, boundary, gx=0.0, gy=-9.81, device=0):
        self.L = hip_lib()
        self.h = C.c_void_p()
        fluid = np.ascontiguousarray(fluid, PARTICLE)
        boundary = np.ascontiguousarray(boundary, PARTICLE)
        self.n, self.nb = len(fluid), len(boundary)
        rc = self.L.sph_create(C.byref(self.h), C.byref(prm), fluid.ctypes.data_as(C.c_void_p), self.n,
                               boundary.ctypes.data_as(C.c_void_p), self.nb, gx, gy, device)
        if rc:
            msg = self.L.sph_last_error(self.h).decode() if self.h else "sph_create failed"
            self.close()
            raise SphError(rc, msg)

    def _chk(self, rc):
        if rc:
            raise SphError(rc, self.L.sph_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.sph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def step(self, nsteps=1, gx=0.0, gy=-9.81):
        self._chk(self.L.sph_step(self.h, gx, gy, nsteps))

    def sync(self):
        self._chk(self.L.sph_sync(self.h))

    def read_particles(self):
        out = np.zeros(self.n, PARTICLE)
        self._chk(self.L.sph_read_particles(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def read_accel(self):
        du, dv = np.zeros(self.n, np.float32), np.zeros(self.n, np.float32)
        self._chk(self.L.sph_read_accel(self.h, du.ctypes.data_as(C.c_void_p), dv.ctypes.data_as(C.c_void_p)))
        return du, dv

    def read_boundary(self):
        out = np.zeros(self.nb, PARTICLE)
        self._chk(self.L.sph_read_boundary(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def update_boundary(self, boundary):
        """move the walls: x, y, u, v of every wall particle (original order); psi is kept."""
        boundary = np.ascontiguousarray(boundary, PARTICLE)
        assert len(boundary) == self.nb
        self._chk(self.L.sph_update_boundary(self.h, boundary.ctypes.data_as(C.c_void_p)))

    def set_boundary_velocity(self, u, v):
        """every wall particle moves with (u, v): what the wall viscosity term sees; nothing is re-binned"""
        self._chk(self.L.sph_set_boundary_velocity(self.h, u, v))

    def stats(self):
        a, b = C.c_float(), C.c_float()
        self._chk(self.L.sph_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def grid_dims(self):
        a, b = C.c_int(), C.c_int()
        self._chk(self.L.sph_grid_dims(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def out_of_domain(self):
        return int(self.L.sph_out_of_domain_count(self.h))

    def device_grid(self):
        """rows, cols, cell length of the device's neighbour grid (cell = 2H + skin)."""
        a, b, c = C.c_int(), C.c_int(), C.c_float()
        self._chk(self.L.sph_device_grid(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def check_stats(self):
        """steps in which a particle was beyond skin/2 and the relative-motion check ran."""
        a = C.c_longlong()
        self._chk(self.L.sph_check_stats(self.h, C.byref(a)))
        return a.value

    def set_rebuild_launches(self, one_launch):
        """True: the rebuild chain of a step is one kernel with grid barriers (default of single-GPU contexts)"""
        self._chk(self.L.sph_set_rebuild_launches(self.h, 1 if one_launch else 0))

    def rebuild_launches(self):
        """True: the last step ran its rebuild chain as one launch"""
        return self.L.sph_get_rebuild_launches(self.h) == 1

    def request_rebuild(self):
        """the next step rebuilds the neighbour structure whatever the displacement criterion says"""
        self._chk(self.L.sph_request_rebuild(self.h))

    def rebuild_stats(self):
        """(rebuilds of the neighbour structure since creation, tiles put on the direct path by them)."""
        a, b = C.c_longlong(), C.c_longlong()
        self._chk(self.L.sph_rebuild_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_verification(self, mode):
        """True / False / None (automatic: from 500 000 particles on) — box groups whose displacement boxes have moved more than the
        skin relative to each other are checked particle by particle inside the launch of the density pass (else they ask for the rebuild)."""
        self._chk(self.L.sph_set_verification(self.h, -1 if mode is None else (1 if mode else 0)))

    def set_list_repair(self, mode):
        """True / False / None (automatic: from 4 000 000 particles on) — missing pairs appended to the lists instead of a rebuild."""
        self.L.sph_set_list_repair.argtypes = [C.c_void_p, C.c_int]
        self._chk(self.L.sph_set_list_repair(self.h, -1 if mode is None else (1 if mode else 0)))

    def rebuild_reasons(self):
        """(unverifiable box pairs, verification found a missing pair, drift cap, unused) — requests for a rebuild so far."""
        a = (C.c_longlong * 4)()
        self._chk(self.L.sph_rebuild_reasons(self.h, a))
        return tuple(int(v) for v in a)

    def repair_stats(self):
        """(pairs appended to the lists instead of a rebuild, not staged in reach, no free byte, queue full) so far."""
        a = (C.c_longlong * 4)()
        self.L.sph_repair_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        self._chk(self.L.sph_repair_stats(self.h, a))
        return tuple(int(x) for x in a)

    def verify_stats(self):
        """pairs of box groups verified particle by particle so far (instead of asking for a rebuild)."""
        v = C.c_longlong(0)
        self._chk(self.L.sph_verify_stats(self.h, C.byref(v)))
        return int(v.value)

    def current_skin(self):
        """the skin of the present neighbour lists, as a fraction of 2H (it adapts between skin_min and skin)."""
        s = float(self.L.sph_current_skin(self.h))
        if s < 0:
            raise SphError(SPH_E_HIP, "sph_current_skin failed")
        return s

    def direct_tile_reasons(self):
        """why tiles went to the direct path so far: counts of (pairs, rows, runs / cell table, candidates, window, list length)."""
        a = (C.c_longlong * 7)()
        self._chk(self.L.sph_direct_tile_reasons(self.h, a))
        return tuple(int(x) for x in a)

    def upload_state(self, fluid):
        fluid = np.ascontiguousarray(fluid, PARTICLE)
        assert len(fluid) == self.n
        self._chk(self.L.sph_upload_state(self.h, fluid.ctypes.data_as(C.c_void_p)))

    def upload_accel(self, du, dv):
        du, dv = np.ascontiguousarray(du, np.float32), np.ascontiguousarray(dv, np.float32)
        assert len(du) == self.n and len(dv) == self.n
        self._chk(self.L.sph_upload_accel(self.h, du.ctypes.data_as(C.c_void_p), dv.ctypes.data_as(C.c_void_p)))

    def eval_density(self):
        self._chk(self.L.sph_eval_density(self.h))

    def eval_pressure(self):
        self._chk(self.L.sph_eval_pressure(self.h))

    def eval_accel(self, gx=0.0, gy=-9.81):
        self._chk(self.L.sph_eval_accel(self.h, gx, gy))

    def profile_steps(self, nsteps, gx=0.0, gy=-9.81):
        kt = KernelTimes()
        self._chk(self.L.sph_profile_steps(self.h, gx, gy, nsteps, C.byref(kt)))
        d = {KERNEL_NAMES[k]: kt.ms[k] for k in range(7)}
        d["step"] = kt.step_ms
        d["rebuilds_per_step"] = kt.rebuilds / max(kt.nsteps, 1)
        return d

    def time_kernel(self, name, reps=20):
        """mean ms per launch of an idempotent kernel ('density_eos', 'force_kick', 'build_list'; 'density_spec' = the density
        launch as sph_step issues it, with the criterion's jobs, + a one-thread reset launch), back-to-back launches."""
        ms = C.c_float()
        kid = 9 if name == "density_spec" else KERNEL_NAMES.index(name)      # SPH_K_DENSITY_SPEC (include/sph_diag.h)
        self._chk(self.L.sph_time_kernel(self.h, kid, reps, C.byref(ms)))
        return ms.value

    def set_stream(self, hip_stream):
        self._chk(self.L.sph_set_stream(self.h, C.c_void_p(hip_stream)))

    def set_variant(self, variant):
        self._chk(self.L.sph_set_variant(self.h, variant))

    def device_bytes(self):
        return int(self.L.sph_device_bytes(self.h))

    def render_metaballs(self):
        buf = np.zeros(1024, np.uint8)
        self._chk(self.L.sph_render_metaballs(self.h, buf.ctypes.data_as(C.c_void_p)))
        return buf


from . import slab  # noqa: E402,F401  (host-side slab decomposition: partitioner, transports, runner)


def _device_cell(prm):
    """cell length of the device grid (2H + skin) from the C ABI, so that slab hosts bin exactly like the device."""
    return float(hip_lib().sph_device_cell(C.byref(prm)))


slab.set_cell_fn(_device_cell)
