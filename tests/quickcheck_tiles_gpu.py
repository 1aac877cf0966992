"""Manual (not collected): direct tiles and their reasons on the golden block fixture and on cfg1 / cfg2 states."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
sph = importlib.import_module("pi-sph-fluid_amd")
import orc
from test_gpu_parity import load_golden, particles, boundary_particles
g = load_golden("block.npz")
prm = sph.default_params(tuple(g["box"]))
f = particles(orc, g["state"], np.float32(prm.rho0) * np.float32(prm.vol))
b = boundary_particles(orc, g["boundary_xy"])
with sph.Context(prm, f, b, 0.0, -9.81) as ctx:
    print("block: n", len(f), "grid", ctx.device_grid(), "rebuilds/direct", ctx.rebuild_stats(), "why", ctx.direct_tile_reasons())
for name, steps in (("cfg1", (0, 1000, 3000, 3000)), ("cfg2", (0, 1000, 3000))):
    prm, f, b = sph.scene(name) if name == "cfg1" else sph.dam_break(1)
    with sph.Context(prm, f, b) as ctx:
        for s in steps:
            ctx.step(s); ctx.sync()
            print(name, "after +%d steps: rebuilds/direct" % s, ctx.rebuild_stats(), "why", ctx.direct_tile_reasons(), "stats", ctx.stats())
