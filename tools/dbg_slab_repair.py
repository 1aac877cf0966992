"""tools/dbg_slab_repair.py — slabs with / without verification and list repair against a single context, step by step (what
tests/test_gpu_slab.py::test_slabs_repair_their_lists asserts).  A debugging aid: python tools/dbg_slab_repair.py on the GPU box."""
import sys, os, importlib, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")


def particles(state, m, rho0=1000.0):
    f = np.zeros(len(state), sph.PARTICLE)
    f["x"], f["y"], f["u"], f["v"] = state[:, 0], state[:, 1], state[:, 2], state[:, 3]
    f["m"] = m
    f["rho"] = rho0
    return f



L = sph.hip_lib()
rng = np.random.default_rng(11)
box = (0.0, 16.0, 0.0, 16.0)
prm = sph.default_params(box)
nx, ny = 168, 40
gx, gy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
xy = np.array([1.7, 6.5]) + 0.075 * np.stack([gx.ravel(), gy.ravel()], 1) + rng.uniform(-0.02, 0.02, (nx * ny, 2))
inner = ((gx.ravel() >= 24) & (gx.ravel() < 60)) | ((gx.ravel() >= 108) & (gx.ravel() < 144))
uv = rng.uniform(-40.0, 40.0, (nx * ny, 2)) * ((rng.random(nx * ny) < 0.2) & inner)[:, None]
state = np.concatenate([xy, uv], 1).astype(np.float32)
f = particles(state, np.float32(prm.rho0) * np.float32(prm.vol))
_p, _f, walls = sph.scene_disc(box, 8.0, 8.0, 0.1)
prm.skin = prm.skin_min = 0.3
refs = {}
for rep in (True, False):
    with sph.Context(prm, f, walls, 0.0, 0.0) as ctx:
        ctx.set_verification(True); ctx.set_list_repair(rep)
        out = []
        for k in (5, 12):
            ctx.step(k - (out[-1][0] if out else 0), 0.0, 0.0); ctx.sync()
            out.append((k, ctx.read_particles()))
        refs[rep] = out
        print("single repair", rep, ctx.repair_stats(), ctx.rebuild_stats())
for k, a in refs[True]:
    b = dict(refs[False])[k]
    print("single repair vs not @%d: dx %.3e drho %.3e" % (k, max(np.abs(a["x"]-b["x"]).max(), np.abs(a["y"]-b["y"]).max()), np.max(np.abs(a["rho"]-b["rho"])/b["rho"])))
for verify, repair in ((0, 0), (1, 0), (1, 1)):
    parts = sph.slab.partition_columns(prm, f, 2, slack=8)
    slabs = [sph.slab.GpuSlab(sph, prm, f, walls, c0, c1, r > 0, r < 1, 0.0, 0.0) for r, (c0, c1) in enumerate(parts)]
    for s_ in slabs:
        assert L.sph_set_verification(s_.h, verify) == 0 and L.sph_set_list_repair(s_.h, repair) == 0
    runner = sph.slab.SlabRunner(slabs, sph.slab.LocalTransport(slabs))
    done = 0
    for k, ref in refs[False]:
        runner.step(k - done, 0.0, 0.0); done = k
        for s_ in slabs: s_.sync()
        out, du, dv, seen = runner.gather_local(len(f), sph.PARTICLE)
        print("slabs verify %d repair %d @%d: seen ok %s dx %.3e drho %.3e" % (verify, repair, k, bool(np.all(seen == 1)), max(np.abs(out["x"]-ref["x"]).max(), np.abs(out["y"]-ref["y"]).max()), np.max(np.abs(out["rho"]-ref["rho"])/ref["rho"])))
    tot = 0
    for s_ in slabs:
        a = (C.c_longlong * 4)(); L.sph_repair_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]; L.sph_repair_stats(s_.h, a); tot += a[0]
        w = (C.c_longlong * 4)(); L.sph_rebuild_reasons.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]; L.sph_rebuild_reasons(s_.h, w)
        v = C.c_longlong(); L.sph_verify_stats(s_.h, C.byref(v))
        print("   slab repairs", list(a), "rebuilds", s_.rebuilds(), "requests", list(w), "verified", v.value)
    for s_ in slabs: s_.close()
