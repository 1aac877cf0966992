"""Manual (not collected): 30 000 steps of cfg2 and of cfg1 with the default parameters: rates, statistics, rebuild and direct-tile
counters, the skin, every 5000 steps; all particles finite and inside the box at the end.  cfg2 runs through (8.0k steps/s
sustained, no direct tile).  cfg1 is the reference's physics beyond its limits after the impact (step ~5500): single particles
leave at 100-900 m/s (the exact walk, variant 1, shows the same), force a rebuild every third step and finally leave the
domain: the run then ends with SPH_E_OUT_OF_DOMAIN, as it should."""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
sph = importlib.import_module("pi-sph-fluid_amd")
import numpy as np
for cfg, name in ((1, "cfg2"), (0, "cfg1")):
    if cfg == 1: prm, f, b = sph.dam_break(1)
    else: prm, f, b = sph.scene("cfg1")
    ctx = sph.Context(prm, f, b)
    t0 = time.perf_counter()
    for k in range(6):
        ctx.step(5000); ctx.sync()
        mr, ms = ctx.stats()
        print(name, "steps", (k + 1) * 5000, "rate %.0f" % ((k + 1) * 5000 / (time.perf_counter() - t0)), "max_rho %.1f max_speed %.1f" % (mr, ms),
              "rebuilds/direct", ctx.rebuild_stats(), "skin %.3f" % ctx.current_skin(), "why", ctx.direct_tile_reasons(), flush=True)
    p = ctx.read_particles()
    assert np.all(np.isfinite(p["x"])) and np.all(np.isfinite(p["rho"]))
    assert p["x"].min() >= prm.x_min and p["x"].max() <= prm.x_max and p["y"].min() >= prm.y_min, (p["x"].min(), p["x"].max(), p["y"].min(), p["y"].max())
    print(name, "ok: x in [%.2f, %.2f], y in [%.2f, %.2f]" % (p["x"].min(), p["x"].max(), p["y"].min(), p["y"].max()), flush=True)
    del ctx
