#!/usr/bin/env python3
"""GPU side of oracle/rho_gate_chaos.py: the distribution of "worst max-rho-error line of a 4000-step run of the default scene" over
perturbed initial positions, for a given build of the HIP library — the bisect the round-5 advisor asked for (is it the v_rcp_f32
EOS, the lower skin floor, or chaos that put one deterministic run at 1.22 %?).

    python tools/rho_gate_gpu.py [lib=csrc/libsph_hip.so] [runs=48] [skin_min=default] [deterministic=0]
    e.g.  make -C pi-sph-fluid_amd variant NAME=ieee VFLAGS=-DSPH_EOS_IEEE
          python tools/rho_gate_gpu.py pi-sph-fluid_amd/csrc/libsph_hip_ieee.so 48
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sph = importlib.import_module("pi-sph-fluid_amd")
from test_gpu_health import stat_lines, perturbed  # noqa: E402


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else sph.LIB_HIP
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    skin_min = float(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "default" else None
    det = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    sph.LIB_HIP = os.path.abspath(lib)
    prm, f, b = sph.scene("cfg0")
    if skin_min is not None:
        prm.skin_min = skin_min
    prm.deterministic = det
    rho0 = np.float32(prm.rho0)
    rng = np.random.default_rng(11)
    worst, second = [], []
    for r in range(runs):
        g = perturbed(f, rng) if r else f
        with sph.Context(prm, g, b, 0.0, -9.81) as ctx:
            lines = stat_lines(lambda k: ctx.step(k, 0.0, -9.81), lambda: float((np.float32(ctx.stats()[0]) - rho0) / rho0 * 100), prm.dt)
            ctx.sync()
        s = sorted(lines)
        worst.append(s[-1])
        second.append(s[-2])
    w, s2 = np.array(worst), np.array(second)
    print(json.dumps({"lib": os.path.basename(lib), "runs": runs, "skin_min": skin_min, "deterministic": det, "run0_worst": float(w[0]),
                      "worst_line": {"min": float(w.min()), "median": float(np.median(w)), "p90": float(np.percentile(w, 90)), "max": float(w.max()),
                                     "over_1pct": int((w > 1).sum())},
                      "second_worst": {"median": float(np.median(s2)), "max": float(s2.max())}}))


if __name__ == "__main__":
    main()
