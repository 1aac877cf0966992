"""Manual GPU sanity run (not collected by pytest): product vs oracle on a few scenes, both kernel variants,
plus first timings.  Usage: python tests/quickcheck_gpu.py [--big]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402

sph = importlib.import_module("pi-sph-fluid_amd")


def to_orc_params(O, prm):
    return O.params((prm.x_min, prm.x_max, prm.y_min, prm.y_max))


def compare(tag, prm, fluid, boundary, variant, steps=0):
    O = orc.Oracle("strict")
    op = to_orc_params(O, prm)
    of, ob = fluid.copy(), boundary.copy()
    O.psi(op, ob)
    du, dv, sa = O.eval(op, of, ob, 0.0, -9.81, want_sum_abs=True)
    ctx = sph.Context(prm, fluid, boundary, 0.0, -9.81)
    ctx.set_variant(variant)
    ctx.upload_state(fluid); ctx.eval_density(); ctx.eval_pressure(); ctx.eval_accel(0.0, -9.81)
    gb = ctx.read_boundary()
    gf = ctx.read_particles()
    gdu, gdv = ctx.read_accel()
    e_psi = np.max(np.abs(gb["m"] - ob["m"]) / ob["m"]) if len(ob) else 0
    e_rho = np.max(np.abs(gf["rho"] - of["rho"]) / of["rho"])
    B = 22857142.0
    e_p = np.max(np.abs(gf["p"] - of["p"]) / (of["p"] + B))
    da = np.hypot(gdu - du, gdv - dv)
    e_a = np.max(da / (sa + 9.81))
    print("%s v%d: n=%d psi %.2e rho %.2e p %.2e a(G3-scale, but from own rho) %.2e  max|da| %.3g" %
          (tag, variant, len(fluid), e_psi, e_rho, e_p, e_a, da.max()))
    # staged G3: upload oracle's rho,p
    ctx.upload_state(of)
    ctx.eval_accel(0.0, -9.81)
    gdu, gdv = ctx.read_accel()
    da = np.hypot(gdu - du, gdv - dv)
    print("   staged G3: max |da|/(sum|terms|+g) = %.2e" % np.max(da / (sa + 9.81)))
    if steps:
        ctx2 = sph.Context(prm, fluid, boundary, 0.0, -9.81)
        ctx2.set_variant(variant)
        ctx2.step(steps)
        ctx2.sync()
        g2 = ctx2.read_particles()
        O.steps(op, of, ob, 0.0, -9.81, du, dv, steps)
        print("   after %d steps: max|dx| = %.3e  max|du| = %.3e" %
              (steps, max(np.abs(g2["x"] - of["x"]).max(), np.abs(g2["y"] - of["y"]).max()),
               max(np.abs(g2["u"] - of["u"]).max(), np.abs(g2["v"] - of["v"]).max())))
        ctx2.close()
    ctx.close()


def timing(tag, prm, fluid, boundary, variant, steps=200):
    t0 = time.time()
    ctx = sph.Context(prm, fluid, boundary, 0.0, -9.81)
    ctx.set_variant(variant)
    t1 = time.time()
    ctx.step(20); ctx.sync()
    t2 = time.time()
    ctx.step(steps); ctx.sync()
    t3 = time.time()
    kt = ctx.profile_steps(20)
    mr, ms = ctx.stats()
    print("%s v%d: n=%d create %.2fs  %.1f steps/s (%.1f Mp-steps/s)  max_rho %.1f max_speed %.2f" %
          (tag, variant, len(fluid), t1 - t0, steps / (t3 - t2), steps / (t3 - t2) * len(fluid) / 1e6, mr, ms))
    print("   kernels ms:", {k: round(v, 4) for k, v in kt.items()})
    ctx.close()


if __name__ == "__main__":
    prm, f, b = sph.scene("cfg0")
    for v in (1, 0):
        compare("cfg0", prm, f, b, v, steps=100)
    prm, f, b = sph.scene_block((0.0, 40.0, 0.0, 8.0), 0.3, 0.3, 240, 60)
    for v in (1, 0):
        compare("block14k", prm, f, b, v, steps=50)
    for name in (["cfg1", "cfg2"] if "--big" in sys.argv else []):
        prm, f, b = sph.scene(name)
        for v in (1, 0):
            timing(name, prm, f, b, v)
