#!/usr/bin/env python3
"""gen_golden.py — TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/*.npz + manifest.json.

Runs only where /root/reference exists: every expected output below is produced by the REAL
reference's own functions (oracle/_ref/libpisph_ref_strict.so, built by oracle/build_ref.sh from
/root/reference/pi_sph_fluid.c, `-O2`, IEEE f32).  The reference has no tests or golden vectors of
its own, so these are the pins.  The as-shipped `-Ofast -march=native` build of the
same source is run beside it and its deviation from the strict build is recorded in the manifest:
that is the reference's own self-consistency and the justification of the staged tolerances of
SURVEY.md §8c.

Fixtures (all f32 little-endian inside .npz):
  drop.npz      the default scene (269 fluid + 162 boundary) at k = 0,1,10,100,1000,2000,4000 steps:
                state (x,y,u,v), psi, rho, p, du, dv, metaball bitmap
  block.npz     240 x 60 dam-break block in a 40 x 8 m box after 3000 steps (developed flow, wall
                contact, p > 0 for most particles): state + rho, p, du, dv
  gas.npz       6400 uniformly random particles with random velocities in a 6 x 6 m box:
                state + rho, p, du, dv  (binning / cutoff edge cases)
"""
import hashlib
import importlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import orc  # noqa: E402

sph = importlib.import_module("pi-sph-fluid_amd")   # host-side scene generators only (no GPU needed)
OUT = os.path.join(ROOT, "tests", "golden")
GX, GY = 0.0, -9.81


def state_of(f):
    return np.stack([f["x"], f["y"], f["u"], f["v"]], axis=1).astype(np.float32)


def outputs(R, fluid, boundary, box):
    """rho, p, du, dv from the reference for the state in `fluid` (boundary carries psi)."""
    f = fluid.copy()
    du, dv = R.eval(f, boundary, box, GX, GY, flags=7, threads=4)
    return f["rho"].copy(), f["p"].copy(), du, dv


def adev(du, dv, edu, edv):
    """acceleration deviation figures of gate G4: max |da| and rms|da| / rms|a|."""
    da = np.hypot(du.astype(np.float64) - edu, dv.astype(np.float64) - edv)
    rms_a = np.sqrt(np.mean(edu.astype(np.float64) ** 2 + edv.astype(np.float64) ** 2))
    return {"a_abs": float(da.max()), "a_rms_rel": float(np.sqrt(np.mean(da ** 2)) / rms_a)}


def dev(a, b):
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    d = np.abs(a64 - b64)
    return {"max_abs": float(d.max()), "max_rel": float((d / np.maximum(np.abs(a64), 1e-30)).max())}


def main():
    subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")])
    R, RF = orc.Reference("strict"), orc.Reference("fast")
    os.makedirs(OUT, exist_ok=True)
    manifest = {"generator": "oracle/gen_golden.py",
                "reference": "/root/reference/pi_sph_fluid.c (hot path :10-411, scene :476-540, step :612-644)",
                "strict_build": "gcc -O2 (no -march, no fast-math)", "fast_build": "gcc -Ofast -march=native (Makefile:2-4)",
                "gcc": subprocess.check_output(["gcc", "--version"]).decode().splitlines()[0],
                "gravity": [GX, GY], "fixtures": {}}

    # ---------------- drop: the default scene ----------------
    box = R.box((0.0, 4.0, 0.0, 2.0))
    f, b = R.scene()
    R.psi(b, box)
    data = {"fluid_xy0": np.stack([f["x"], f["y"]], 1), "boundary_xy": np.stack([b["x"], b["y"]], 1),
            "psi": b["m"].copy(), "constants": R.constants()}
    ks = [0, 1, 10, 100, 1000, 2000, 4000]
    du, dv = R.eval(f, b, box, GX, GY, flags=7, threads=4)
    ff, bf = RF.scene()
    RF.psi(bf, box)
    duf, dvf = RF.eval(ff, bf, box, GX, GY, flags=7, threads=4)
    self_dev = {"psi": dev(b["m"], bf["m"])}
    k_now = 0
    maxn = (0, 0)
    for k in ks:
        if k > k_now:
            R.steps(f, b, box, GX, GY, du, dv, k - k_now, threads=4)
            RF.steps(ff, bf, box, GX, GY, duf, dvf, k - k_now, threads=4)
            k_now = k
        data["state_%d" % k] = state_of(f)
        data["rho_%d" % k], data["p_%d" % k] = f["rho"].copy(), f["p"].copy()
        # du/dv: the accelerations the loop itself holds after step k (computed from the half-kicked
        # velocities, :632) — needed to continue the trajectory; eval_du/eval_dv: calculate_accelerations
        # evaluated on the stored state — the expected output of the staged gates
        data["du_%d" % k], data["dv_%d" % k] = du.copy(), dv.copy()
        _, _, data["eval_du_%d" % k], data["eval_dv_%d" % k] = outputs(R, f, b, box)
        data["metaballs_%d" % k] = R.metaballs(f, box)
        mn = R.max_neighbors(f, b, box)
        maxn = (max(maxn[0], mn[0]), max(maxn[1], mn[1]))
        # the fast build evaluated on the STRICT state: pure arithmetic deviation, no trajectory divergence
        g = f.copy()
        gdu, gdv = RF.eval(g, b, box, GX, GY, flags=7, threads=4)
        self_dev["k%d" % k] = {"rho": dev(f["rho"], g["rho"]), "p": dev(f["p"], g["p"]),
                               **adev(gdu, gdv, data["eval_du_%d" % k], data["eval_dv_%d" % k]),
                               "traj_dx_fast_vs_strict": float(max(np.abs(ff["x"] - f["x"]).max(), np.abs(ff["y"] - f["y"]).max()))}
    assert maxn[0] <= 48 and maxn[1] <= 48, maxn
    np.savez_compressed(os.path.join(OUT, "drop.npz"), **data)
    manifest["fixtures"]["drop.npz"] = {"n_fluid": len(f), "n_boundary": len(b), "box": [0, 4, 0, 2], "steps": ks,
                                        "max_neighbors": maxn, "fast_vs_strict": self_dev,
                                        "sum_psi": float(b["m"].astype(np.float64).sum()),
                                        "sum_rho_0": float(data["rho_0"].astype(np.float64).sum())}

    # ---------------- block: developed dam break ----------------
    bx = (0.0, 40.0, 0.0, 8.0)
    prm, f, b = sph.scene_block(bx, 0.3, 0.3, 240, 60)
    f, b = f.view(orc.PARTICLE).copy(), b.view(orc.PARTICLE).copy()
    box = R.box(bx)
    R.psi(b, box)
    xy0 = np.stack([f["x"], f["y"]], 1)
    du, dv = R.eval(f, b, box, GX, GY, flags=7, threads=8)
    nsteps = 3000
    R.steps(f, b, box, GX, GY, du, dv, nsteps, threads=8)
    mn = R.max_neighbors(f, b, box)
    assert mn[0] <= 48 and mn[1] <= 48, mn
    g = f.copy()
    gdu, gdv = RF.eval(g, b, box, GX, GY, flags=7, threads=8)
    _, _, edu, edv = outputs(R, f, b, box)
    np.savez_compressed(os.path.join(OUT, "block.npz"), fluid_xy0=xy0, boundary_xy=np.stack([b["x"], b["y"]], 1),
                        psi=b["m"].copy(), state=state_of(f), rho=f["rho"].copy(), p=f["p"].copy(), du=du, dv=dv,
                        eval_du=edu, eval_dv=edv,
                        box=np.array(bx, np.float32), nsteps=np.int32(nsteps))
    manifest["fixtures"]["block.npz"] = {"n_fluid": len(f), "n_boundary": len(b), "box": list(bx), "steps": nsteps,
                                         "max_neighbors": mn, "p_positive_fraction": float((f["p"] > 0).mean()),
                                         "rho_range": [float(f["rho"].min()), float(f["rho"].max())],
                                         "p_max": float(f["p"].max()),
                                         "speed_max": float(np.hypot(f["u"], f["v"]).max()),
                                         "fast_vs_strict": {"rho": dev(f["rho"], g["rho"]), "p": dev(f["p"], g["p"]),
                                                            **adev(gdu, gdv, edu, edv)}}

    # ---------------- gas: random positions / velocities ----------------
    bx = (0.0, 6.0, 0.0, 6.0)
    seed = 20231003
    rng = np.random.default_rng(seed)
    n = 6400
    prm = sph.default_params(bx)
    _, _, b = sph.scene_block(bx, 0.3, 0.3, 1, 1)
    b = b.view(orc.PARTICLE).copy()
    f = np.zeros(n, orc.PARTICLE)
    f["x"] = rng.uniform(0.05, 5.95, n).astype(np.float32)
    f["y"] = rng.uniform(0.05, 5.95, n).astype(np.float32)
    f["u"] = rng.uniform(-2, 2, n).astype(np.float32)
    f["v"] = rng.uniform(-2, 2, n).astype(np.float32)
    f["m"] = np.float32(prm.rho0) * np.float32(prm.vol)
    f["rho"] = prm.rho0
    box = R.box(bx)
    R.psi(b, box)
    mn = R.max_neighbors(f, b, box)
    assert mn[0] <= 48 and mn[1] <= 48, mn
    rho, p, du, dv = outputs(R, f, b, box)
    g = f.copy()
    RF.psi(b.copy(), box)
    gdu, gdv = RF.eval(g, b, box, GX, GY, flags=7, threads=4)
    gas_self = {"rho": dev(rho, g["rho"]), "p": dev(p, g["p"]), **adev(gdu, gdv, du, dv)}
    np.savez_compressed(os.path.join(OUT, "gas.npz"), boundary_xy=np.stack([b["x"], b["y"]], 1), psi=b["m"].copy(),
                        state=state_of(f), rho=rho, p=p, eval_du=du, eval_dv=dv, box=np.array(bx, np.float32))
    manifest["fixtures"]["gas.npz"] = {"n_fluid": n, "n_boundary": len(b), "box": list(bx), "seed": seed,
                                       "max_neighbors": mn, "rho_range": [float(rho.min()), float(rho.max())],
                                       "fast_vs_strict": gas_self}

    for name in list(manifest["fixtures"]):
        with open(os.path.join(OUT, name), "rb") as fh:
            manifest["fixtures"][name]["sha256"] = hashlib.sha256(fh.read()).hexdigest()
            manifest["fixtures"][name]["bytes"] = os.path.getsize(os.path.join(OUT, name))
    with open(os.path.join(OUT, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)
    print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
