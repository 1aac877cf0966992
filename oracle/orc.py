"""ctypes bindings for the CPU oracle (oracle/sph_oracle.c) and, when present,
the compiled real reference (oracle/_ref, see oracle/build_ref.sh).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# byte-compatible with the reference's `struct particle` (pi_sph_fluid.c:26-31)
PARTICLE = np.dtype([("x", "<f4"), ("y", "<f4"), ("u", "<f4"), ("v", "<f4"),
                     ("m", "<f4"), ("rho", "<f4"), ("p", "<f4")])
assert PARTICLE.itemsize == 28


class OrcParams(C.Structure):
    _fields_ = [("r", C.c_float), ("h", C.c_float), ("rho0", C.c_float), ("c", C.c_float),
                ("g", C.c_float), ("dt", C.c_float), ("vol", C.c_float),
                ("x_min", C.c_float), ("x_max", C.c_float), ("y_min", C.c_float), ("y_max", C.c_float),
                ("alpha", C.c_double), ("eps", C.c_double), ("k1", C.c_double), ("k2", C.c_double)]


class RefBox(C.Structure):
    _fields_ = [("x_min", C.c_float), ("x_max", C.c_float), ("y_min", C.c_float), ("y_max", C.c_float)]


def build(fast=True, strict=True):
    """Compile the oracle (and the reference harness when /root/reference exists)."""
    targets = []
    if strict:
        targets.append("liborc_strict.so")
    if fast:
        targets.append("liborc_fast.so")
    subprocess.check_call(["make", "-s", "-C", HERE] + targets)
    subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """The CPU restatement. kind = 'strict' (-O2, bit-exact gate) or 'fast' (reference flags, timing)."""

    def __init__(self, kind="strict"):
        san = os.environ.get("ORC_SANITIZE", "")      # "asan": the ASan + UBSan build (make -C oracle asan; tests/test_sanitizers.py)
        path = os.path.join(HERE, "liborc_%s%s.so" % (kind, "_" + san if san else ""))
        if san and not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", HERE, san])
        if not os.path.exists(path):
            build(fast=(kind == "fast"), strict=(kind == "strict"))
        self.lib = C.CDLL(path)
        self.kind = kind
        L = self.lib
        L.orc_params_default.argtypes = [C.POINTER(OrcParams)]
        L.orc_constants.argtypes = [C.POINTER(OrcParams), C.c_void_p]
        L.orc_grid_dims.argtypes = [C.POINTER(OrcParams), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_scene_default.argtypes = [C.POINTER(OrcParams), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_first_touch_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int]
        L.orc_first_touch_copy.restype = None
        L.orc_psi.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_int]
        L.orc_max_neighbors.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_eval.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                               C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_steps.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_metaballs.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_int, C.c_int]

    def params(self, box=None):
        p = OrcParams()
        self.lib.orc_params_default(C.byref(p))
        if box is not None:
            p.x_min, p.x_max, p.y_min, p.y_max = [float(v) for v in box]
        return p

    def constants(self, p):
        out = np.zeros(16, np.float32)
        self.lib.orc_constants(C.byref(p), _ptr(out))
        return out

    def grid_dims(self, p):
        n, m = C.c_int(), C.c_int()
        self.lib.orc_grid_dims(C.byref(p), C.byref(n), C.byref(m))
        return n.value, m.value

    def scene_default(self, p):
        pf, pb, nf, nb = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        self.lib.orc_scene_default(C.byref(p), C.byref(pf), C.byref(nf), C.byref(pb), C.byref(nb))
        f = np.ctypeslib.as_array(C.cast(pf, C.POINTER(C.c_float)), (nf.value * 7,)).copy().view(PARTICLE)
        b = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_float)), (nb.value * 7,)).copy().view(PARTICLE)
        self.lib.orc_free(pf)
        self.lib.orc_free(pb)
        return f, b

    def psi(self, p, boundary):
        rc = self.lib.orc_psi(C.byref(p), _ptr(boundary), len(boundary))
        if rc:
            raise RuntimeError("orc_psi rc=%d" % rc)
        return boundary

    def max_neighbors(self, p, fluid, boundary):
        a, b = C.c_int(), C.c_int()
        rc = self.lib.orc_max_neighbors(C.byref(p), _ptr(fluid), len(fluid), _ptr(boundary), len(boundary),
                                        C.byref(a), C.byref(b))
        if rc:
            raise RuntimeError("orc_max_neighbors rc=%d" % rc)
        return a.value, b.value

    def eval(self, p, fluid, boundary, gx, gy, flags=7, threads=0, want_sum_abs=False):
        """In-place on fluid (rho, p); returns (du, dv[, sum_abs])."""
        n = len(fluid)
        du, dv = np.zeros(n, np.float32), np.zeros(n, np.float32)
        sa = np.zeros(n, np.float32)
        if want_sum_abs:
            flags |= 8
        rc = self.lib.orc_eval(C.byref(p), _ptr(fluid), n, _ptr(boundary), len(boundary), gx, gy, flags,
                               _ptr(du), _ptr(dv), _ptr(sa), threads or (os.cpu_count() or 1))
        if rc:
            raise RuntimeError("orc_eval rc=%d" % rc)
        return (du, dv, sa) if want_sum_abs else (du, dv)

    def steps(self, p, fluid, boundary, gx, gy, du, dv, nsteps, threads=0):
        rc = self.lib.orc_steps(C.byref(p), _ptr(fluid), len(fluid), _ptr(boundary), len(boundary), gx, gy,
                                _ptr(du), _ptr(dv), nsteps, threads or (os.cpu_count() or 1))
        if rc:
            raise RuntimeError("orc_steps rc=%d" % rc)

    def first_touch(self, a, threads):
        """a copy of `a` whose pages were first touched by the threads that will work on them (timing runs)"""
        out = np.empty_like(a)
        self.lib.orc_first_touch_copy(_ptr(out), _ptr(np.ascontiguousarray(a)), len(a), a.dtype.itemsize, threads)
        return out

    def metaballs(self, p, fluid, threads=0):
        buf = np.zeros(1024, np.uint8)
        rc = self.lib.orc_metaballs(C.byref(p), _ptr(buf), _ptr(fluid), len(fluid), threads or (os.cpu_count() or 1))
        if rc:
            raise RuntimeError("orc_metaballs rc=%d" % rc)
        return buf


class Reference:
    """The real reference's hot path compiled by oracle/build_ref.sh (N <= 65 534, H fixed)."""

    def __init__(self, kind="strict"):
        path = os.path.join(HERE, "_ref", "libpisph_ref_%s.so" % kind)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        L = self.lib
        L.ref_constants.argtypes = [C.c_void_p]
        L.ref_grid_dims.argtypes = [C.POINTER(RefBox), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_psi.argtypes = [C.c_void_p, C.c_int, C.POINTER(RefBox)]
        L.ref_max_neighbors.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(RefBox),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_eval.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(RefBox), C.c_float, C.c_float,
                               C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_steps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(RefBox), C.c_float, C.c_float,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ref_scene.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_metaballs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(RefBox), C.c_int]

    @staticmethod
    def available(kind="strict"):
        return os.path.exists(os.path.join(HERE, "_ref", "libpisph_ref_%s.so" % kind))

    @staticmethod
    def box(b=(0.0, 4.0, 0.0, 2.0)):
        return RefBox(*[float(v) for v in b])

    def constants(self):
        out = np.zeros(16, np.float32)
        self.lib.ref_constants(_ptr(out))
        return out

    def grid_dims(self, box):
        n, m = C.c_int(), C.c_int()
        self.lib.ref_grid_dims(C.byref(box), C.byref(n), C.byref(m))
        return n.value, m.value

    def scene(self):
        pf, pb, nf, nb = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        self.lib.ref_scene(C.byref(pf), C.byref(nf), C.byref(pb), C.byref(nb))
        f = np.ctypeslib.as_array(C.cast(pf, C.POINTER(C.c_float)), (nf.value * 7,)).copy().view(PARTICLE)
        b = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_float)), (nb.value * 7,)).copy().view(PARTICLE)
        self.lib.ref_free(pf)
        self.lib.ref_free(pb)
        # the reference leaves .p of fluid and .m/.p of boundary uninitialised (:502, :526); zero them
        f["p"] = 0
        b["m"] = 0
        b["p"] = 0
        return f, b

    def psi(self, boundary, box):
        if self.lib.ref_psi(_ptr(boundary), len(boundary), C.byref(box)):
            raise RuntimeError("ref_psi: too many particles for the reference")
        return boundary

    def max_neighbors(self, fluid, boundary, box):
        a, b = C.c_int(), C.c_int()
        if self.lib.ref_max_neighbors(_ptr(fluid), len(fluid), _ptr(boundary), len(boundary), C.byref(box),
                                      C.byref(a), C.byref(b)):
            raise RuntimeError("ref_max_neighbors failed")
        return a.value, b.value

    def eval(self, fluid, boundary, box, gx, gy, flags=7, threads=4):
        n = len(fluid)
        du, dv = np.zeros(n, np.float32), np.zeros(n, np.float32)
        if self.lib.ref_eval(_ptr(fluid), n, _ptr(boundary), len(boundary), C.byref(box), gx, gy, flags,
                             _ptr(du), _ptr(dv), threads):
            raise RuntimeError("ref_eval: too many particles for the reference")
        return du, dv

    def steps(self, fluid, boundary, box, gx, gy, du, dv, nsteps, threads=4):
        if self.lib.ref_steps(_ptr(fluid), len(fluid), _ptr(boundary), len(boundary), C.byref(box), gx, gy,
                              _ptr(du), _ptr(dv), nsteps, threads):
            raise RuntimeError("ref_steps: too many particles for the reference")

    def metaballs(self, fluid, box, threads=4):
        buf = np.zeros(1024, np.uint8)
        if self.lib.ref_metaballs(_ptr(buf), _ptr(fluid), len(fluid), C.byref(box), threads):
            raise RuntimeError("ref_metaballs failed")
        return buf
