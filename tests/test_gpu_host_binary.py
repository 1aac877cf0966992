"""Row f3: the C host program (pi-sph-fluid_amd/host/desktop_sph_fluid.c, the `desktop_sph_fluid` of the reference's
Makefile:18-23 with every physics call of main() replaced by a C-ABI call) is RUN, on the default scene, and what it
prints and produces is checked:

  * the start-up lines of pi_sph_fluid.c:543-545 (dt / expected ticks/s, n_fluid = 269, n_boundary = 162),
  * the statistics line of :683-687 every 0.1 s of simulated time (format and content),
  * the frame of the final state against the reference's draw_metaballs bitmap (golden) — at step 1000, where
    trajectories still agree to 1e-3 m, pixel by pixel; at step 4000 (decorrelated) as lit area / overlap,
  * the final fluid[] against the golden trajectory (step 1000) and its aggregates (step 4000),
  * a run under the scripted tilt trace (sph_gravity, the MPU6050 stand-in) against the oracle stepping under the same
    gravity samples.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu

HOST = os.path.join(ROOT, "pi-sph-fluid_amd", "host", "desktop_sph_fluid")
STAT = re.compile(r"^sim time: (\d+\.\d\d), ticks/s: (-?\d+), max rho error: (-?\d+\.\d{3})% \(worst\) (-?\d+\.\d{3})%, "
                  r"max speed: (\d+\.\d) m/s \(worst\) (\d+\.\d) m/s, $")


def run_host(tmp_path, *args):
    frame, state = str(tmp_path / "frame.bin"), str(tmp_path / "state.bin")
    r = subprocess.run([HOST, "--dump-frame", frame, "--dump-state", state] + list(args), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode().splitlines()
    fr = np.fromfile(frame, np.uint8)
    st = np.fromfile(state, np.float32).reshape(-1, 7)
    assert len(fr) == 1024
    return out, fr, st


def test_default_scene_console_and_frame(sph, tmp_path):
    out, frame, st = run_host(tmp_path, "--scene", "cfg0", "--steps", "1000")
    assert out[0] == "dt = 0.000244    (expected ticks/s) 4102"          # :543 (probe output of the reference)
    assert out[1] == "n_fluid = 269" and out[2] == "n_boundary = 162"    # :544-545
    stats = [STAT.match(ln) for ln in out if ln.startswith("sim time")]
    assert len(stats) == 2 and all(stats)                                # 1000 steps = 0.244 s: lines at 0.10 and 0.20 s
    assert [m.group(1) for m in stats] == ["0.10", "0.20"]
    assert all(int(m.group(2)) > 0 for m in stats)                       # ticks/s
    assert 9.81 * 0.2 - 0.1 <= float(stats[1].group(5)) <= 9.81 * 0.2 + 0.7    # free fall g t, plus the drop's own expansion
    g = load_golden("drop.npz")
    ref = g["state_1000"]
    assert len(st) == 269
    assert max(np.abs(st[:, 0] - ref[:, 0]).max(), np.abs(st[:, 1] - ref[:, 1]).max()) <= 1e-3      # G5 @ 1000
    assert np.max(np.abs(st[:, 5] - g["rho_1000"]) / g["rho_1000"]) <= 5e-3
    assert np.count_nonzero(np.unpackbits(frame) != np.unpackbits(g["metaballs_1000"])) <= 4
    done = [ln for ln in out if ln.startswith("done:")]
    assert len(done) == 1 and "1000 steps" in done[0]


def test_default_scene_4000_steps_aggregates(sph, tmp_path):
    # (--deterministic: a chaotic 4000-step run then gives the same bits every time, so these aggregate checks cannot flicker
    # with the arrival order of the sort's atomics)
    out, frame, st = run_host(tmp_path, "--scene", "cfg0", "--steps", "4000", "--show", "--deterministic")
    g = load_golden("drop.npz")
    stats = [STAT.match(ln) for ln in out if ln.startswith("sim time")]
    assert len(stats) == 9 and all(stats)                                # 0.975 s of simulated time
    worst_speed = max(float(m.group(6)) for m in stats)
    assert 4.0 <= worst_speed <= 8.0                                     # the drop hits the floor at ~4.3 m/s and splashes
    rho_err = [float(m.group(3)) for m in stats]
    # the author's health criterion: ~1 % (:16, :662).  The statistic is the maximum over the particles at one instant of a chaotic
    # splash, and this deterministic run is ONE trajectory of it: the reference's own arithmetic from initial positions perturbed by
    # +-4 ulp gives a worst line between 0.10 and 1.36 % (oracle/rho_gate_chaos.py, 600 runs: median 0.25 %, 4 runs beyond 1 %, the
    # second-worst line never beyond 0.50 %), so one trajectory is held to the reference's RANGE — at most one line may spike, none
    # beyond 2 % — and the arithmetic itself to the reference's DISTRIBUTION by tests/test_gpu_health.py (24 perturbed runs; round 6,
    # with the bisect the round-5 advisor asked for: IEEE divisions in the EOS, skin_min 0.12 and the deterministic order all give the
    # same distribution, tools/rho_gate_gpu.py)
    assert max(rho_err) < 2.0 and sorted(rho_err)[-2] < 1.0, rho_err
    ref = g["state_4000"]
    assert abs(st[:, 1].mean() - ref[:, 1].mean()) <= 0.05 * ref[:, 1].mean()
    assert st[:, 0].min() > 0 and st[:, 0].max() < 4 and st[:, 1].min() > 0.09 and st[:, 1].max() < 2
    assert abs(st[:, 5].max() - g["rho_4000"].max()) <= 0.02 * g["rho_4000"].max()
    got, exp = np.unpackbits(frame).astype(bool), np.unpackbits(g["metaballs_4000"]).astype(bool)
    assert abs(int(got.sum()) - int(exp.sum())) <= 0.1 * exp.sum()       # same amount of fluid on the panel ...
    assert (got & exp).sum() / (got | exp).sum() >= 0.7                  # ... in the same place (a pool on the floor)
    # --show printed text frames (two pixel rows per text row) with fluid in them
    assert sum(1 for ln in out if len(ln) == 128 and "#" in ln) > 10


def test_tilt_trace_against_oracle(sph, orc, oracle, tmp_path):
    steps = 500
    out, frame, st = run_host(tmp_path, "--scene", "cfg0", "--steps", str(steps), "--tilt", "--tilt-amp", "20",
                              "--tilt-period", "0.1", "--tilt-hold", "0.004")
    prm, f, b = sph.scene("cfg0")
    p = oracle.params()
    of, ob = f.view(orc.PARTICLE).copy(), b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    grav = sph.GravitySource(sph.GRAVITY_TILT, 9.81, amp_deg=20.0, period_s=0.1, hold_s=0.004)
    gx, gy = grav.sample(0.0)
    du, dv = oracle.eval(p, of, ob, gx, gy, threads=4)
    t = np.float32(0.0)
    seen = set()
    for _ in range(steps):               # the host's order: step under the current sample, advance t, re-sample (:632, :678)
        oracle.steps(p, of, ob, gx, gy, du, dv, 1, threads=4)
        t = np.float32(t + np.float32(prm.dt))
        gx, gy = grav.sample(float(t))
        seen.add((gx, gy))
    assert len(seen) > 20                                                # the trace really varied
    assert max(np.abs(st[:, 0] - of["x"]).max(), np.abs(st[:, 1] - of["y"]).max()) <= 1e-4
    assert np.abs(st[:, 0] - f["x"]).max() > 1e-2                        # and pushed the drop sideways
