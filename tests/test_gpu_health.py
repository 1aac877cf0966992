"""The reference author's health criterion — "max rho error" of the statistics line, about 1 % (pi_sph_fluid.c:16, :657-687) — as a
property of the ARITHMETIC and not of one trajectory.

The statistic is the maximum over 269 particles at nine instants of a chaotic splash.  The reference's own arithmetic, stepped
from initial positions that differ by a few ulp (oracle/rho_gate_chaos.py: the CPU oracle, bit-identical to the -O2 reference;
600 runs at +-4 ulp), gives for "the worst of a run's nine lines": median 0.25 %, p75 0.36 %, p90 0.45 %, p99 0.85 %, max 1.36 %;
4 runs of 600 beyond 1 %, 19 beyond 0.6 %; second-worst line at most 0.50 %.  One GPU trajectory beyond 1 % therefore says nothing
(round 5's deterministic run happened to be such a one: 1.22 %), and one below 1 % says as little.  What a regression of the
arithmetic would move is the DISTRIBUTION: this test steps 24 perturbed copies of the default scene through the C ABI (the GPU's
summation order differs from run to run: every execution draws 24 fresh trajectories) and holds their distribution to the
reference's, with bounds that a sample of 24 from the reference's distribution breaks about once in 5000 executions — P(6 or more
of 24 beyond 0.6 %) ~ 1e-4, P(4 or more beyond 1 %) ~ 2e-5, the median of 24 has a standard deviation of ~0.04 % — and that a
doubling of the typical error, or a tail three times as heavy, does not pass.  (tools/rho_gate_gpu.py, 64 runs per build: the shipped
build, IEEE divisions in the EOS, skin_min 0.12 and the deterministic order all give medians 0.22-0.24 % and 0-1 runs beyond 1 %.)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RUNS, ULPS, STEPS = 24, 4, 4000


def stat_lines(step_fn, rho_max_fn, dt, steps=STEPS):
    """the instants of the host's statistics line (t - last_t > 0.1f, f32 time: pi_sph_fluid.c:678-691)"""
    t, last_t, lines, todo = np.float32(0.0), np.float32(0.0), [], 0
    for _ in range(steps):
        todo += 1
        t = np.float32(t + np.float32(dt))
        if np.float32(t - last_t) > np.float32(0.1):
            step_fn(todo)
            todo = 0
            lines.append(rho_max_fn())
            last_t = t
    if todo:
        step_fn(todo)
    return lines


def perturbed(f, rng, ulps=ULPS):
    g = f.copy()
    for k in ("x", "y"):
        g[k] = (g[k].view(np.int32) + rng.integers(-ulps, ulps + 1, len(g)).astype(np.int32)).view(np.float32)
    return g


def test_density_health_statistic_has_the_references_distribution(sph, orc, oracle):
    prm, f, b = sph.scene("cfg0")
    rho0 = np.float32(prm.rho0)
    rng = np.random.default_rng(11)
    worst, second = [], []
    for r in range(RUNS):
        g = perturbed(f, rng) if r else f
        with sph.Context(prm, g, b, 0.0, -9.81) as ctx:
            lines = stat_lines(lambda k: ctx.step(k, 0.0, -9.81), lambda: float((np.float32(ctx.stats()[0]) - rho0) / rho0 * 100), prm.dt)
            ctx.sync()
        assert len(lines) == 9
        s = sorted(lines)
        worst.append(s[-1])
        second.append(s[-2])
    worst, second = np.array(worst), np.array(second)
    # the reference (oracle/rho_gate_chaos.py, 600 runs): median 0.25, p75 0.36, p90 0.45, p99 0.85, max 1.36; 3.2 % of the runs beyond 0.6 %, 0.67 % beyond 1 %
    assert np.median(worst) < 0.40, worst
    assert (worst > 0.6).sum() <= 5, worst
    assert (worst > 1.0).sum() <= 3, worst
    assert worst.max() < 2.5, worst
    assert np.median(second) < 0.30 and second.max() < 1.0, second
    # ... and the oracle on three of the same perturbed scenes, here and now
    p = oracle.params()
    ob = b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    rng = np.random.default_rng(11)
    ow = []
    for r in range(3):
        g = (perturbed(f, rng) if r else f).view(orc.PARTICLE).copy()
        du, dv = oracle.eval(p, g, ob, 0.0, -9.81, threads=4)
        lines = stat_lines(lambda k: oracle.steps(p, g, ob, 0.0, -9.81, du, dv, k, threads=4),
                           lambda: float((g["rho"].max() - rho0) / rho0 * 100), prm.dt)
        ow.append(max(lines))
    assert max(ow) < 2.0 and min(ow) > 0.05, ow                      # (the checker computes the same statistic on these scenes, now)
    assert abs(np.median(worst) - 0.25) < 0.15, worst                # the GPU's median against the reference's (600 runs: 0.25 %)
