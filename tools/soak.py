"""tools/soak.py [steps [every [scene]]] — cfg2 (or cfg4 under its tilt trace) for `steps` steps (default 12 000); every `every` (default 500): rho from the live neighbour
lists against the exact walk over the cell ranges (variant 1) on the same state, conservation, flags.  (GPU box.)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 500
scene = sys.argv[3] if len(sys.argv) > 3 else "cfg2"      # cfg4: 32 M particles under the scripted tilt trace (list repair on by default there)
prm, f, b = sph.dam_break(1) if scene == "cfg2" else sph.scene(scene)
grav = sph.GravitySource(sph.GRAVITY_TILT, 9.81) if scene == "cfg4" else None
dt_sim, sim = float(np.float32(prm.dt)), 0
worst = 0.0
with sph.Context(prm, f, b, 0.0, -9.81) as ctx:
    for k in range(every, steps + 1, every):
        if grav is None:
            ctx.step(every, 0.0, -9.81)
        else:
            for _ in range(every):      # (the reference re-reads g every step: pi_sph_fluid.c:632)
                gx, gy = grav.sample(sim * dt_sim)
                ctx.step(1, gx, gy)
                sim += 1
        ctx.sync()
        ctx.set_variant(0)
        ctx.eval_density()
        rho_list = ctx.read_particles()["rho"]
        ctx.set_variant(1)
        ctx.eval_density()
        ref = ctx.read_particles()
        ctx.set_variant(0)
        ctx.eval_density()
        ctx.eval_pressure()
        err = float(np.max(np.abs(rho_list - ref["rho"]) / ref["rho"]))
        worst = max(worst, err)
        print("step %6d: max |rho_list - rho_exact| / rho = %.2e, rebuilds %d, verified pairs %d, reasons %s, repairs %s, skin %.3f, oob %d, |v|max %.1f"
              % (k, err, ctx.rebuild_stats()[0], ctx.verify_stats(), ctx.rebuild_reasons(), ctx.repair_stats(), ctx.current_skin(), ctx.out_of_domain(),
                 float(np.hypot(ref["u"], ref["v"]).max())), flush=True)
        assert np.all(np.isfinite(ref["x"])) and err <= 4e-6, err
print("ok: worst %.2e over %d steps" % (worst, steps))
