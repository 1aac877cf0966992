"""Manual (not collected): 40 000 steps of cfg1 (the 262 144-particle drop, impact at t = 1.4 s) with the reused neighbour
lists checked against an exact walk of the live state every 4000 steps.  The splash reaches > 100 m/s (beyond the
reference's c/10 design limit) and eventually leaves the single-layer box: SPH_E_OUT_OF_DOMAIN ends the run, as the
reference's heap overflow would (pi_sph_fluid.c:111-116)."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
sph = importlib.import_module("pi-sph-fluid_amd")
prm, f, b = sph.scene("cfg1")
ctx = sph.Context(prm, f, b)
t0 = time.time()
for k in range(10):
    ctx.step(4000)
    try:
        ctx.sync()
    except sph.SphError as e:
        print("stopped:", e)
        break
    p = ctx.read_particles()
    # lists vs exact walk on the live state
    ctx.set_variant(0); ctx.eval_density(); ra = ctx.read_particles()["rho"]
    ctx.set_variant(1); ctx.eval_density(); rb = ctx.read_particles()["rho"]
    ctx.set_variant(0)
    mr, ms = ctx.stats()
    print("step %6d  t=%.2fs  max_rho %.1f max_speed %.2f  y_min %.3f  rebuilds %s checks %d  lists-vs-walk %.2e  finite %s" % (
        (k + 1) * 4000, (k + 1) * 4000 * prm.dt, mr, ms, p["y"].min(), ctx.rebuild_stats(), ctx.check_stats(),
        np.max(np.abs(ra - rb) / rb), bool(np.all(np.isfinite(p["x"])))), flush=True)
print("wall %.1fs" % (time.time() - t0))
