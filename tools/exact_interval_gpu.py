"""Manual (not collected): how long would the neighbour lists of the developed dam break REALLY last?  Positions P0 at a
rebuild, then after k steps: the largest reference distance |r(P0)| among the pairs that are inside the support 2H now.  Lists
built at P0 with skin s hold every pair with |r(P0)| < 2H + s: they are complete as long as that maximum stays below 2H + s.
Compared with the steps at which the device's (conservative, box-based) criterion asks for a rebuild."""
import importlib, os, sys, time
import numpy as np
from scipy.spatial import cKDTree
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
prm, f, b = sph.dam_break(1)
two_h = 2 * float(prm.h)
ctx = sph.Context(prm, f, b)
ctx.step(warm); ctx.sync()
ctx.request_rebuild(); ctx.step(1); ctx.sync()
r0 = ctx.rebuild_stats()[0]
p0 = ctx.read_particles()
P0 = np.stack([p0["x"], p0["y"]], 1).astype(np.float64)
done = 0
for k in (2, 4, 6, 8, 12, 16, 24, 32):
    ctx.step(k - done); ctx.sync(); done = k
    pk = ctx.read_particles()
    Pk = np.stack([pk["x"], pk["y"]], 1).astype(np.float64)
    t0 = time.time()
    pairs = cKDTree(Pk).query_pairs(two_h, output_type="ndarray")
    dref = np.linalg.norm(P0[pairs[:, 0]] - P0[pairs[:, 1]], axis=1)
    ex = (dref - two_h) / two_h
    print("k=%2d: %d pairs inside 2H now; max (|r_ref| - 2H)/2H = %.3f, 99.999 pct %.3f; pairs beyond skin 0.15: %d, 0.20: %d, 0.30: %d | device rebuilds since P0: %d | max speed %.1f (%.0f s)" % (
        k, len(pairs), ex.max(), np.percentile(ex, 99.999), int((ex > 0.15).sum()), int((ex > 0.20).sum()), int((ex > 0.30).sum()),
        ctx.rebuild_stats()[0] - r0, float(np.hypot(pk["u"], pk["v"]).max()), time.time() - t0), flush=True)
