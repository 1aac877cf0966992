#!/usr/bin/env python3
"""How much does the reference's OWN health statistic — "max rho error" of the line it prints every 0.1 s of simulated time,
pi_sph_fluid.c:657-687 — scatter over runs of the default scene that differ by rounding only?

The CPU oracle (bit-identical to the -O2 reference) steps the default scene 4000 times from initial positions perturbed by a few
ulp (what a different summation order does to a trajectory within a few hundred steps), and the nine statistics lines of each run
are collected the way the host program collects them (t - last_t > 0.1f).  The spread of "the worst line of a run" is what the
gate of tests/test_gpu_host_binary.py::test_default_scene_4000_steps_aggregates has to allow for: a GPU run is one more such
trajectory.  CPU only; ~2 s per run.

    python oracle/rho_gate_chaos.py [runs=24] [ulps=4]      -> table + JSON line
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402  (this script is test infrastructure like gen_golden.py: it produces the thresholds of tests/test_gpu_health.py)


def run(O, p, f0, b, steps=4000, threads=4):
    f = f0.copy()
    du, dv = O.eval(p, f, b, 0.0, -9.81, threads=threads)
    t = np.float32(0.0)
    last_t = np.float32(0.0)
    lines = []
    for _ in range(steps):
        O.steps(p, f, b, 0.0, -9.81, du, dv, 1, threads=threads)
        t = np.float32(t + np.float32(p.dt))
        if np.float32(t - last_t) > np.float32(0.1):
            lines.append(float((f["rho"].max() - np.float32(p.rho0)) / np.float32(p.rho0) * 100))
            last_t = t
    return lines


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    ulps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    O = orc.Oracle("strict")
    p = O.params()
    f, b = O.scene_default(p)
    O.psi(p, b)
    rng = np.random.default_rng(7)
    rows = []
    for r in range(runs):
        g = f.copy()
        if r:      # run 0: the reference's own trajectory
            for k in ("x", "y"):
                bits = g[k].view(np.int32) + rng.integers(-ulps, ulps + 1, len(g)).astype(np.int32)
                g[k] = bits.view(np.float32)
        ln = run(O, p, g, b)
        rows.append(ln)
        s = sorted(ln)
        print("run %2d: worst %.3f %%  second %.3f %%  lines %s" % (r, s[-1], s[-2], " ".join("%.2f" % v for v in ln)), flush=True)
    worst = np.array([max(ln) for ln in rows])
    second = np.array([sorted(ln)[-2] for ln in rows])
    out = {"runs": runs, "ulps": ulps, "lines_per_run": len(rows[0]),
           "worst_line_of_a_run_pct": {"min": float(worst.min()), "median": float(np.median(worst)), "p75": float(np.percentile(worst, 75)),
                                       "p90": float(np.percentile(worst, 90)), "p99": float(np.percentile(worst, 99)), "max": float(worst.max()),
                                       "runs_over_1pct": int((worst > 1.0).sum()), "runs_over_0.6pct": int((worst > 0.6).sum())},
           "second_worst_line_pct": {"median": float(np.median(second)), "max": float(second.max())}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
