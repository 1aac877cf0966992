"""tools/slab_repair_explore.py — which scene makes a SLAB repair its lists (tests/test_gpu_slab.py::test_slabs_repair_their_lists needs one
where it must)?  The jittered lattice cut into two slabs; `n_fast` particles wholly inside each slab's interior fly at `speed` m/s (random
directions), everybody else is at rest.  Prints, per configuration and per checkpoint, the slabs' repairs / rebuilds / requests.  (GPU box.)
    python tools/slab_repair_explore.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")
L = sph.hip_lib()
L.sph_repair_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
L.sph_rebuild_reasons.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]


def scene(n_fast, speed, jitter=0.02, seed=11):
    rng = np.random.default_rng(seed)
    box = (0.0, 16.0, 0.0, 16.0)
    prm = sph.default_params(box)
    nx, ny = 168, 40
    gx, gy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    gx, gy = gx.ravel(), gy.ravel()
    xy = np.array([1.7, 6.5]) + 0.075 * np.stack([gx, gy], 1) + (rng.uniform(-jitter, jitter, (nx * ny, 2)) if jitter > 0 else 0.0)
    uv = np.zeros((nx * ny, 2))
    for lo, hi in ((30, 54), (114, 138)):      # the middle of either slab, well inside vertically
        cand = np.nonzero((gx >= lo) & (gx < hi) & (gy >= 10) & (gy < 30))[0]
        pick = rng.choice(cand, n_fast, replace=False)
        ang = rng.uniform(0, 2 * np.pi, n_fast)
        uv[pick] = speed * np.stack([np.cos(ang), np.sin(ang)], 1)
    f = np.zeros(nx * ny, sph.PARTICLE)
    f["x"], f["y"], f["u"], f["v"] = xy[:, 0], xy[:, 1], uv[:, 0], uv[:, 1]
    f["m"] = np.float32(prm.rho0) * np.float32(prm.vol)
    f["rho"] = prm.rho0
    _p, _f, walls = sph.scene_disc(box, 8.0, 8.0, 0.1)
    prm.skin = prm.skin_min = 0.3
    return prm, f, walls


def run_config(n_fast, speed, jitter):
    prm, f, walls = scene(n_fast, speed, jitter)
    for repair in (1, 0):
            parts = sph.slab.partition_columns(prm, f, 2, slack=8)
            slabs = [sph.slab.GpuSlab(sph, prm, f, walls, c0, c1, r > 0, r < 1, 0.0, 0.0) for r, (c0, c1) in enumerate(parts)]
            for s_ in slabs:
                assert L.sph_set_verification(s_.h, 1) == 0 and L.sph_set_list_repair(s_.h, repair) == 0
            runner = sph.slab.SlabRunner(slabs, sph.slab.LocalTransport(slabs))
            line = []
            for upto in (8, 16, 24, 32):
                runner.step(8, 0.0, 0.0)
                for s_ in slabs:
                    s_.sync()
                reps, reqs = [], []
                for s_ in slabs:
                    a = (C.c_longlong * 4)()
                    L.sph_repair_stats(s_.h, a)
                    w = (C.c_longlong * 4)()
                    L.sph_rebuild_reasons(s_.h, w)
                    reps.append(list(a))
                    reqs.append(list(w))
                line.append("@%d repairs %s rebuilds %d requests %s" % (upto, [r[0] for r in reps], slabs[0].rebuilds(), reqs))
            print("n_fast %d speed %.0f jitter %.4f repair %d: %s" % (n_fast, speed, jitter, repair, " | ".join(line)), flush=True)
            for s_ in slabs:
                s_.close()


for cfg in ((12, 20.0, 0.002), (1, 20.0, 0.002), (40, 30.0, 0.002), (12, 20.0, 0.0005), (12, 20.0, 0.0)):
    try:
        run_config(*cfg)
    except Exception as e:      # (reported: the next configuration still runs)
        print("config", cfg, "failed:", repr(e)[:300], flush=True)
